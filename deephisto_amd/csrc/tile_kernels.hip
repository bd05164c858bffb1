// Tile-side kernels of the deephisto hot path for MI355X (gfx950, wave64):
//   a1  dh_tile_grid            host integer restatement of the dense grid order
//   --  dh_synth_slide          closed-form benchmark slide generated in HBM
//   a2/a4/a5 dh_tile_gather     uint8 HWC slide -> [n,P,P,3] / [n,3,P,P] f32|bf16, exactly k/255
//   a8  dh_accumulate_logits    ordered (bit-exact) canvas accumulation, dh_argmax_map
// All of these are HBM-bound byte/integer work: no LDS, no MFMA; the design
// rules are full-width coalesced rows (one wave = one tile row) and 16-byte stores.
#include <errno.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "dh_common.h"
#include "env_knobs.h"

namespace dh {
static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}
}  // namespace dh

namespace dh {
namespace {
struct EnvKnob { const char* name; long def, lo, hi; const char* read; const char* effect; };
#define DH_ENV_ROW(n, d, l, h, r, e) {#n, (long)(d), (long)(l), (long)(h), r, e},
const EnvKnob g_env_knobs[] = {DH_ENV_KNOBS(DH_ENV_ROW)};
#undef DH_ENV_ROW
// 0: unset (default), 1: valid (*out), -1: set but not a decimal integer in [lo, hi]
int env_parse(const EnvKnob& k, long* out) {
  const char* v = getenv(k.name);
  *out = k.def;
  if (!v) return 0;
  char* end = nullptr;
  errno = 0;
  const long x = strtol(v, &end, 10);
  if (end == v || *end != '\0' || errno != 0 || x < k.lo || x > k.hi) return -1;
  *out = x;
  return 1;
}
void env_complain(const EnvKnob& k) {
  set_error("%s='%s' is not an integer in [%ld, %ld] (default %ld): %s", k.name, getenv(k.name), k.lo, k.hi, k.def, k.effect);
}
}  // namespace

int env_int(const char* name) {
  for (const EnvKnob& k : g_env_knobs)
    if (!strcmp(k.name, name)) {
      long v;
      if (env_parse(k, &v) < 0) {
        env_complain(k);
        fprintf(stderr, "libdeephisto_hip: %s -- default used\n", g_err.c_str());
      }
      return (int)v;
    }
  fprintf(stderr, "libdeephisto_hip: internal error: environment knob %s is not in env_knobs.h\n", name);
  abort();
}

int env_check() {
  for (const EnvKnob& k : g_env_knobs) {
    long v;
    if (env_parse(k, &v) < 0) { env_complain(k); return DH_EINVAL; }
  }
  return DH_OK;
}
}  // namespace dh

// test hook: the table of env_knobs.h as text, one "NAME default lo hi read" line per knob (INTEGRATION.md carries the same table)
extern "C" int dh_debug_env_knobs(char* buf, int64_t cap) {
  std::string s;
  char line[256];
  for (const dh::EnvKnob& k : dh::g_env_knobs) {
    snprintf(line, sizeof line, "%s %ld %ld %ld %s\n", k.name, k.def, k.lo, k.hi, k.read);
    s += line;
  }
  DH_REQUIRE(buf && (int64_t)s.size() + 1 <= cap, "dh_debug_env_knobs: buffer of %lld bytes too small (%zu needed)", (long long)cap, s.size() + 1);
  memcpy(buf, s.c_str(), s.size() + 1);
  return DH_OK;
}

extern "C" int dh_abi_version(void) { return 1; }
extern "C" const char* dh_last_error(void) { return dh::g_err.c_str(); }

// ---------------------------------------------------------------------------
// a1: tile grid (host).  Reference: patch_samplers/full_samplers.py:374-404.
// ---------------------------------------------------------------------------
static inline int64_t range_len(int64_t stop, int64_t step) {  // len(range(0, stop, step))
  return stop <= 0 ? 0 : (stop + step - 1) / step;
}

extern "C" int dh_tile_grid_count(int64_t h, int64_t w, int32_t patch, int32_t stride,
                                  int32_t batch, int64_t* n_unique, int64_t* n_padded) {
  DH_REQUIRE(patch > 0 && stride > 0 && batch > 0, "tile grid: patch, stride, batch must be > 0");
  DH_REQUIRE(h >= patch && w >= patch, "tile grid: slide %lldx%lld smaller than patch %d",
             (long long)h, (long long)w, patch);
  DH_REQUIRE(h <= INT32_MAX && w <= INT32_MAX, "tile grid: slide side exceeds int32");
  const int64_t ny = range_len(h - patch, stride), nx = range_len(w - patch, stride);
  const int64_t n = ny * nx + ny + nx + 1;
  if (n_unique) *n_unique = n;
  if (n_padded) *n_padded = (n + batch - 1) / batch * batch;
  return DH_OK;
}

extern "C" int dh_tile_grid(int64_t h, int64_t w, int32_t patch, int32_t stride, int32_t batch,
                            int32_t* out, int64_t capacity_pairs) {
  int64_t n = 0, np = 0;
  int rc = dh_tile_grid_count(h, w, patch, stride, batch, &n, &np);
  if (rc) return rc;
  DH_REQUIRE(out != nullptr, "tile grid: null output");
  DH_REQUIRE(capacity_pairs >= np, "tile grid: capacity %lld < %lld padded origins",
             (long long)capacity_pairs, (long long)np);
  const int64_t ylim = h - patch, xlim = w - patch;
  int32_t* p = out;
  for (int64_t y = 0; y < ylim; y += stride)
    for (int64_t x = 0; x < xlim; x += stride) { *p++ = (int32_t)y; *p++ = (int32_t)x; }
  for (int64_t y = 0; y < ylim; y += stride) { *p++ = (int32_t)y; *p++ = (int32_t)xlim; }
  for (int64_t x = 0; x < xlim; x += stride) { *p++ = (int32_t)ylim; *p++ = (int32_t)x; }
  for (int64_t i = n - 1; i < np; ++i) {  // the corner, then its padding copies
    *p++ = (int32_t)ylim; *p++ = (int32_t)xlim;
  }
  return DH_OK;
}

// ---------------------------------------------------------------------------
// synthetic slide: pix(y,x,c,seed) = (((y*K_Y)^(x*K_X)^(c*K_C)^(seed*K_S)) >> 7) & 0xFF
// One thread = 16 consecutive bytes of the HWC image (one 16-B store).
// ---------------------------------------------------------------------------
#define DH_KY 73856093u
#define DH_KX 19349663u
#define DH_KC 83492791u
#define DH_KS 2654435761u

__global__ __launch_bounds__(256) void synth_slide_kernel(uint8_t* __restrict__ out, int64_t total,
                                                          uint32_t row_bytes, uint32_t seed_term) {
  const int64_t chunks = (total + 15) >> 4;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < chunks;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i0 = t << 4;
    uint32_t y = (uint32_t)(i0 / row_bytes);
    uint32_t xb = (uint32_t)(i0 - (int64_t)y * row_bytes);
    uint32_t x = xb / 3u, c = xb - 3u * x;
    uint32_t wv[4] = {0, 0, 0, 0};
    const int nb = (total - i0 >= 16) ? 16 : (int)(total - i0);
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const uint32_t v = (((y * DH_KY) ^ (x * DH_KX) ^ (c * DH_KC) ^ seed_term) >> 7) & 0xFFu;
      wv[b >> 2] |= v << (8 * (b & 3));
      if (++c == 3u) { c = 0; ++x; }
      if (++xb == row_bytes) { xb = 0; x = 0; c = 0; ++y; }
    }
    if (nb == 16) {
      *reinterpret_cast<uint4*>(out + i0) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
    } else {
      for (int b = 0; b < nb; ++b) out[i0 + b] = (uint8_t)(wv[b >> 2] >> (8 * (b & 3)));
    }
  }
}

extern "C" int dh_synth_slide(uint8_t* slide, int64_t h, int64_t w, uint32_t seed, void* stream) {
  DH_REQUIRE(slide && h > 0 && w > 0, "synth slide: bad arguments");
  DH_REQUIRE(w * 3 <= (int64_t)UINT32_MAX, "synth slide: row too long");
  DH_REQUIRE((reinterpret_cast<uintptr_t>(slide) & 15) == 0, "synth slide: base must be 16-B aligned");
  const int64_t total = h * w * 3;
  const int64_t chunks = (total + 15) >> 4;
  const int grid = (int)std::min<int64_t>((chunks + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(synth_slide_kernel, dim3(grid), dim3(256), 0, dh::as_stream(stream), slide,
                     total, (uint32_t)(w * 3), seed * DH_KS);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

// ---------------------------------------------------------------------------
// a2/a4/a5: gather.  k/255 in float32, correctly rounded, without a divide:
//   q = k*r; e = fma(-q,255,k); q' = fma(e,r,q)   with r = RN(1/255).
// (tests/test_div255.py checks all 256 inputs against IEEE division on the host;
//  tests/test_gpu_tiles.py checks the kernel's outputs bit-for-bit.)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float div255(uint32_t k) {
  const float r = 1.0f / 255.0f;
  const float kf = (float)k;
  const float q = kf * r;
  const float e = __builtin_fmaf(-q, 255.0f, kf);
  return __builtin_fmaf(e, r, q);
}

__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {  // RNE; inputs are finite
  const uint32_t u = __float_as_uint(f);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

struct __attribute__((packed, aligned(1))) Px4 { uint32_t a, b, c; };   // 4 RGB pixels
struct __attribute__((packed, aligned(1))) U32u { uint32_t v; };

constexpr int GATHER_ROWS = 16;  // tile rows per 256-thread block (4 per wave)

// The gathered tiles are written once and read by a later kernel (0.8 GB per 1 024 tiles: far past the 256 MB Infinity Cache), the slide
// bytes are read once: non-temporal accesses keep neither in the caches.  Measured (tools/hbm_mix_peak.hip: the same 1 : 4 byte mix as a
// linear stream reaches 5.1-5.4 TB/s plain, 5.4-5.5 TB/s non-temporal; profiles/r04_exp_tiler.txt for the kernels).
#ifndef DH_TILER_NT
#define DH_TILER_NT 1
#endif
typedef float f4v_t __attribute__((ext_vector_type(4)));
typedef uint32_t u2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_f4(float* p, float a, float b, float c, float d) {
  if (DH_TILER_NT) __builtin_nontemporal_store(f4v_t{a, b, c, d}, reinterpret_cast<f4v_t*>(p));
  else *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
__device__ __forceinline__ void st_u2(uint16_t* p, uint32_t x, uint32_t y) {
  if (DH_TILER_NT) __builtin_nontemporal_store(u2v_t{x, y}, reinterpret_cast<u2v_t*>(p));
  else *reinterpret_cast<uint2*>(p) = make_uint2(x, y);
}

// NCHW: one wave = one tile row; lane j owns pixels 4j..4j+3 (12 contiguous bytes) and
// writes one 16-B (f32) / 8-B (bf16) store into each of the three planes.
template <bool BF16>
__global__ __launch_bounds__(256) void gather_nchw_kernel(const uint8_t* __restrict__ slide,
                                                          int64_t row_bytes,
                                                          const int32_t* __restrict__ yx, int P,
                                                          void* __restrict__ outv) {
  const int t = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int y0 = yx[2 * t], x0 = yx[2 * t + 1];
  const int r0 = blockIdx.x * GATHER_ROWS + wave;
  const int64_t plane = (int64_t)P * P;
  for (int pj = lane * 4; pj < P; pj += 256) {
#pragma unroll
    for (int rr = 0; rr < GATHER_ROWS / 4; ++rr) {
      const int r = r0 + rr * 4;
      if (r < P) {
        const uint8_t* src = slide + (int64_t)(y0 + r) * row_bytes + (int64_t)(x0 + pj) * 3;
        const Px4 v = *reinterpret_cast<const Px4*>(src);
        const uint32_t b[3] = {v.a, v.b, v.c};
        float f[3][4];
#pragma unroll
        for (int i = 0; i < 12; ++i) f[i % 3][i / 3] = div255((b[i >> 2] >> (8 * (i & 3))) & 0xFFu);
        const int64_t o = (int64_t)t * 3 * plane + (int64_t)r * P + pj;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if constexpr (BF16) {
            st_u2(static_cast<uint16_t*>(outv) + o + c * plane, f32_to_bf16_bits(f[c][0]) | (f32_to_bf16_bits(f[c][1]) << 16),
                  f32_to_bf16_bits(f[c][2]) | (f32_to_bf16_bits(f[c][3]) << 16));
          } else {
            st_f4(static_cast<float*>(outv) + o + c * plane, f[c][0], f[c][1], f[c][2], f[c][3]);
          }
        }
      }
    }
  }
}

// NHWC: the tile row is 3P contiguous bytes -> 3P contiguous outputs; lane j owns
// bytes 4j..4j+3 of each 256-byte piece (4-B load, 16-B / 8-B store).
template <bool BF16>
__global__ __launch_bounds__(256) void gather_nhwc_kernel(const uint8_t* __restrict__ slide,
                                                          int64_t row_bytes,
                                                          const int32_t* __restrict__ yx, int P,
                                                          void* __restrict__ outv) {
  const int t = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int y0 = yx[2 * t], x0 = yx[2 * t + 1];
  const int r0 = blockIdx.x * GATHER_ROWS + wave;
  const int rowlen = 3 * P;  // multiple of 4 (P % 4 == 0)
  for (int bj = lane * 4; bj < rowlen; bj += 256) {
#pragma unroll
    for (int rr = 0; rr < GATHER_ROWS / 4; ++rr) {
      const int r = r0 + rr * 4;
      if (r < P) {
        const uint8_t* src = slide + (int64_t)(y0 + r) * row_bytes + (int64_t)x0 * 3 + bj;
        const uint32_t v = reinterpret_cast<const U32u*>(src)->v;
        const float f0 = div255(v & 0xFFu), f1 = div255((v >> 8) & 0xFFu),
                    f2 = div255((v >> 16) & 0xFFu), f3 = div255(v >> 24);
        const int64_t o = ((int64_t)t * P + r) * rowlen + bj;
        if constexpr (BF16) {
          st_u2(static_cast<uint16_t*>(outv) + o, f32_to_bf16_bits(f0) | (f32_to_bf16_bits(f1) << 16), f32_to_bf16_bits(f2) | (f32_to_bf16_bits(f3) << 16));
        } else {
          st_f4(static_cast<float*>(outv) + o, f0, f1, f2, f3);
        }
      }
    }
  }
}

// Any patch size (P % 4 != 0): one thread per output element.
template <bool BF16, bool NCHW>
__global__ __launch_bounds__(256) void gather_generic_kernel(const uint8_t* __restrict__ slide,
                                                             int64_t row_bytes,
                                                             const int32_t* __restrict__ yx, int P,
                                                             void* __restrict__ outv) {
  const int t = blockIdx.y;
  const int y0 = yx[2 * t], x0 = yx[2 * t + 1];
  const int64_t per_tile = (int64_t)P * P * 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_tile;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / (3 * P));
    const int rem = (int)(i - (int64_t)r * 3 * P);
    const int px = rem / 3, c = rem - 3 * px;
    const float f = div255(slide[(int64_t)(y0 + r) * row_bytes + (int64_t)(x0 + px) * 3 + c]);
    const int64_t o = (int64_t)t * per_tile + (NCHW ? ((int64_t)c * P + r) * P + px : i);
    if constexpr (BF16) static_cast<uint16_t*>(outv)[o] = (uint16_t)f32_to_bf16_bits(f);
    else static_cast<float*>(outv)[o] = f;
  }
}

extern "C" int dh_tile_gather(const uint8_t* slide, int64_t h, int64_t w, const int32_t* yx_dev,
                              const int32_t* yx_host_check, int64_t n, int32_t P, int32_t layout,
                              int32_t dtype, void* out, void* stream) {
  DH_REQUIRE(n >= 0 && n <= 65535, "tile gather: n=%lld out of range [0, 65535]", (long long)n);
  if (n == 0) return DH_OK;  // empty batch: nothing to read or write (pointers may be null)
  DH_REQUIRE(slide && yx_dev && out, "tile gather: null pointer");
  DH_REQUIRE(P > 0 && h >= P && w >= P, "tile gather: patch %d does not fit %lldx%lld", P,
             (long long)h, (long long)w);
  DH_REQUIRE(layout == DH_LAYOUT_NHWC || layout == DH_LAYOUT_NCHW, "tile gather: bad layout %d", layout);
  DH_REQUIRE(dtype == DH_DTYPE_F32 || dtype == DH_DTYPE_BF16, "tile gather: bad dtype %d", dtype);
  if (yx_host_check)
    for (int64_t i = 0; i < n; ++i) {
      const int64_t y = yx_host_check[2 * i], x = yx_host_check[2 * i + 1];
      DH_REQUIRE(y >= 0 && x >= 0 && y + P <= h && x + P <= w,
                 "tile gather: origin %lld = (%lld,%lld) outside slide", (long long)i, (long long)y,
                 (long long)x);
    }
  const int esz = dtype == DH_DTYPE_F32 ? 4 : 2;
  const bool fast = (P % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  (void)esz;
  hipStream_t st = dh::as_stream(stream);
  const int64_t rb = w * 3;
  if (fast) {
    dim3 grid((P + GATHER_ROWS - 1) / GATHER_ROWS, (unsigned)n), block(256);
    if (layout == DH_LAYOUT_NCHW) {
      if (dtype == DH_DTYPE_F32) hipLaunchKernelGGL(gather_nchw_kernel<false>, grid, block, 0, st, slide, rb, yx_dev, P, out);
      else hipLaunchKernelGGL(gather_nchw_kernel<true>, grid, block, 0, st, slide, rb, yx_dev, P, out);
    } else {
      if (dtype == DH_DTYPE_F32) hipLaunchKernelGGL(gather_nhwc_kernel<false>, grid, block, 0, st, slide, rb, yx_dev, P, out);
      else hipLaunchKernelGGL(gather_nhwc_kernel<true>, grid, block, 0, st, slide, rb, yx_dev, P, out);
    }
  } else {
    const int64_t per_tile = (int64_t)P * P * 3;
    dim3 grid((unsigned)std::min<int64_t>((per_tile + 255) / 256, 1024), (unsigned)n), block(256);
    if (layout == DH_LAYOUT_NCHW) {
      if (dtype == DH_DTYPE_F32) hipLaunchKernelGGL((gather_generic_kernel<false, true>), grid, block, 0, st, slide, rb, yx_dev, P, out);
      else hipLaunchKernelGGL((gather_generic_kernel<true, true>), grid, block, 0, st, slide, rb, yx_dev, P, out);
    } else {
      if (dtype == DH_DTYPE_F32) hipLaunchKernelGGL((gather_generic_kernel<false, false>), grid, block, 0, st, slide, rb, yx_dev, P, out);
      else hipLaunchKernelGGL((gather_generic_kernel<true, false>), grid, block, 0, st, slide, rb, yx_dev, P, out);
    }
  }
  DH_LAUNCH_CHECK();
  return DH_OK;
}

// Gather with the batch-level flips of the training pipeline (train.py:71-81 applies
// RandomHorizontalFlip / RandomVerticalFlip to the whole [B,C,H,W] batch: one coin per batch).
template <bool BF16, bool NCHW>
__global__ __launch_bounds__(256) void gather_aug_kernel(const uint8_t* __restrict__ slide, int64_t row_bytes, int h, int w,
                                                         const int32_t* __restrict__ yx, int P, int flip_h, int flip_v,
                                                         void* __restrict__ outv) {
  const int t = blockIdx.y;
  const int y0 = yx[2 * t], x0 = yx[2 * t + 1];
  const int64_t per_tile = (int64_t)P * P * 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_tile; i += (int64_t)gridDim.x * blockDim.x) {
    // i enumerates OUTPUT elements in output order
    int r, px, c;
    if (NCHW) { c = (int)(i / ((int64_t)P * P)); const int rem = (int)(i - (int64_t)c * P * P); r = rem / P; px = rem - r * P; }
    else { r = (int)(i / (3 * P)); const int rem = (int)(i - (int64_t)r * 3 * P); px = rem / 3; c = rem - 3 * px; }
    const int sr = flip_v ? P - 1 - r : r, sx = flip_h ? P - 1 - px : px;
    // region samplers keep the reference's origin bounds, which let a patch hang over the image border
    // (region_samplers.py:112-118): pixels outside the slide read as 0, never as memory outside the allocation
    const int yy = y0 + sr, xx = x0 + sx;
    const bool in = yy >= 0 && yy < h && xx >= 0 && xx < w;
    const float f = in ? div255(slide[(int64_t)yy * row_bytes + (int64_t)xx * 3 + c]) : 0.f;
    const int64_t o = (int64_t)t * per_tile + i;
    if constexpr (BF16) static_cast<uint16_t*>(outv)[o] = (uint16_t)f32_to_bf16_bits(f);
    else static_cast<float*>(outv)[o] = f;
  }
}

extern "C" int dh_tile_gather_aug(const uint8_t* slide, int64_t h, int64_t w, const int32_t* yx_dev, int64_t n, int32_t P,
                                  int32_t layout, int32_t dtype, int32_t flip_h, int32_t flip_v, void* out, void* stream) {
  DH_REQUIRE(n >= 0 && n <= 65535, "tile gather aug: n=%lld out of range", (long long)n);
  if (n == 0) return DH_OK;
  DH_REQUIRE(slide && yx_dev && out && P > 0 && h >= P && w >= P && h <= INT32_MAX && w <= INT32_MAX, "tile gather aug: bad arguments");
  DH_REQUIRE((layout == DH_LAYOUT_NHWC || layout == DH_LAYOUT_NCHW) && (dtype == DH_DTYPE_F32 || dtype == DH_DTYPE_BF16),
             "tile gather aug: bad layout/dtype");
  hipStream_t st = dh::as_stream(stream);
  const int64_t per_tile = (int64_t)P * P * 3;
  dim3 grid((unsigned)std::min<int64_t>((per_tile + 255) / 256, 1024), (unsigned)n), block(256);
  const int64_t rb = w * 3;
  if (layout == DH_LAYOUT_NCHW) {
    if (dtype == DH_DTYPE_F32) hipLaunchKernelGGL((gather_aug_kernel<false, true>), grid, block, 0, st, slide, rb, (int)h, (int)w, yx_dev, P, flip_h, flip_v, out);
    else hipLaunchKernelGGL((gather_aug_kernel<true, true>), grid, block, 0, st, slide, rb, (int)h, (int)w, yx_dev, P, flip_h, flip_v, out);
  } else {
    if (dtype == DH_DTYPE_F32) hipLaunchKernelGGL((gather_aug_kernel<false, false>), grid, block, 0, st, slide, rb, (int)h, (int)w, yx_dev, P, flip_h, flip_v, out);
    else hipLaunchKernelGGL((gather_aug_kernel<true, false>), grid, block, 0, st, slide, rb, (int)h, (int)w, yx_dev, P, flip_h, flip_v, out);
  }
  DH_LAUNCH_CHECK();
  return DH_OK;
}

// NHWC float32 WITHOUT the /255 (FullImageRndSampler.generator_torch, full_samplers.py:286, yields the
// raw 0..255 values as floats -- unlike the dense sampler).
__global__ __launch_bounds__(256) void gather_raw_nhwc_kernel(const uint8_t* __restrict__ slide, int64_t row_bytes, int h, int w,
                                                              const int32_t* __restrict__ yx, int P, float* __restrict__ out) {
  const int t = blockIdx.y;
  const int y0 = yx[2 * t], x0 = yx[2 * t + 1];
  const int64_t per_tile = (int64_t)P * P * 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_tile; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / (3 * P)), rem = (int)(i - (int64_t)r * 3 * P);
    const int yy = y0 + r, xx = x0 + rem / 3;
    const bool in = yy >= 0 && yy < h && xx >= 0 && xx < w;   // outside the slide: 0 (see gather_aug_kernel)
    out[(int64_t)t * per_tile + i] = in ? (float)slide[(int64_t)yy * row_bytes + (int64_t)x0 * 3 + rem] : 0.f;
  }
}

extern "C" int dh_tile_gather_raw(const uint8_t* slide, int64_t h, int64_t w, const int32_t* yx_dev, int64_t n, int32_t P,
                                  float* out, void* stream) {
  DH_REQUIRE(n >= 0 && n <= 65535, "tile gather raw: n=%lld out of range", (long long)n);
  if (n == 0) return DH_OK;
  DH_REQUIRE(slide && yx_dev && out && P > 0 && h >= P && w >= P && h <= INT32_MAX && w <= INT32_MAX, "tile gather raw: bad arguments");
  const int64_t per_tile = (int64_t)P * P * 3;
  dim3 grid((unsigned)std::min<int64_t>((per_tile + 255) / 256, 1024), (unsigned)n), block(256);
  hipLaunchKernelGGL(gather_raw_nhwc_kernel, grid, block, 0, dh::as_stream(stream), slide, w * 3, (int)h, (int)w, yx_dev, P, out);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

__global__ void coords_kernel(const int32_t* __restrict__ yx, int64_t n2, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n2) out[i] = (float)yx[i];
}

extern "C" int dh_tile_coords_f32(const int32_t* yx_dev, int64_t n, float* out, void* stream) {
  if (n == 0) return DH_OK;
  DH_REQUIRE(yx_dev && out && n > 0, "tile coords: bad arguments");
  hipLaunchKernelGGL(coords_kernel, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0,
                     dh::as_stream(stream), yx_dev, 2 * n, out);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

// ---------------------------------------------------------------------------
// a8: ordered accumulation.  The canvas is cut into bins of G x G cells
// (G = max(1, P/d)); the host builds, per bin, the list of tiles that touch it,
// in tile order.  One workgroup owns one bin, so every canvas cell has exactly
// one writer that applies the covering tiles' logits sequentially in the
// reference's order: float32 results are bit-identical to the NumPy `+=` loop,
// with no atomics.  Traffic: canvas read+write once, logits/lists from L2.
// ---------------------------------------------------------------------------
struct BinGeom { int32_t G, bins_x, dh, dw, n_cls, P, d; };

__global__ __launch_bounds__(256) void accumulate_bins_kernel(
    const float* __restrict__ logits, const int32_t* __restrict__ yx,
    const int32_t* __restrict__ bin_start, const int32_t* __restrict__ bin_tiles, BinGeom g,
    float* __restrict__ canvas) {
  __shared__ int32_t s_t[256], s_y0[256], s_y1[256], s_x0[256], s_x1[256];
  const int bin = blockIdx.x;
  const int beg = bin_start[bin], end = bin_start[bin + 1];
  if (beg == end) return;  // uniform for the block
  const int by = bin / g.bins_x, bx = bin - by * g.bins_x;
  const int cells = g.G * g.G, items = cells * g.n_cls;
  for (int it0 = 0; it0 < items; it0 += 256) {
    const int it = it0 + threadIdx.x;
    const int cell = it / g.n_cls, cls = it - cell * g.n_cls;
    const int cy = by * g.G + cell / g.G, cx = bx * g.G + cell % g.G;
    const bool live = it < items && cy < g.dh && cx < g.dw;
    const int64_t addr = ((int64_t)cy * g.dw + cx) * g.n_cls + cls;
    float acc = live ? canvas[addr] : 0.f;
    for (int l0 = beg; l0 < end; l0 += 256) {
      __syncthreads();
      const int l = l0 + threadIdx.x;
      if (l < end) {
        const int t = bin_tiles[l];
        const int y = yx[2 * t], x = yx[2 * t + 1];
        s_t[threadIdx.x] = t;
        s_y0[threadIdx.x] = y / g.d; s_y1[threadIdx.x] = (y + g.P) / g.d;
        s_x0[threadIdx.x] = x / g.d; s_x1[threadIdx.x] = (x + g.P) / g.d;
      }
      __syncthreads();
      const int m = min(256, end - l0);
      if (live)
        for (int k = 0; k < m; ++k)
          if (cy >= s_y0[k] && cy < s_y1[k] && cx >= s_x0[k] && cx < s_x1[k])
            acc = acc + logits[(int64_t)s_t[k] * g.n_cls + cls];
    }
    if (live) canvas[addr] = acc;
  }
}

__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ canvas, int64_t n_cells,
                                                     int n_cls, int64_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cells;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float* p = canvas + i * n_cls;
    float best = p[0];
    int bi = 0;
    for (int c = 1; c < n_cls; ++c) {  // NumPy argmax: first maximum; a NaN wins once
      const float v = p[c];
      if (v > best || (v != v && best == best)) { best = v; bi = c; }
    }
    out[i] = bi;
  }
}

extern "C" int dh_argmax_map(const float* canvas, int64_t n_cells, int32_t n_cls, int64_t* map,
                             void* stream) {
  DH_REQUIRE(canvas && map && n_cells >= 0 && n_cls > 0, "argmax map: bad arguments");
  if (n_cells == 0) return DH_OK;
  const int grid = (int)std::min<int64_t>((n_cells + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(argmax_kernel, dim3(grid), dim3(256), 0, dh::as_stream(stream), canvas, n_cells,
                     n_cls, map);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_accumulate_logits(const float* logits, const int32_t* yx_host, int64_t n,
                                    int32_t P, int32_t d, int32_t n_cls, int64_t h, int64_t w,
                                    float* canvas, int64_t* map, void* stream) {
  DH_REQUIRE(logits && yx_host && canvas, "accumulate: null pointer");
  DH_REQUIRE(P > 0 && d > 0 && n_cls > 0 && h > 0 && w > 0 && n >= 0, "accumulate: bad sizes");
  DH_REQUIRE(n <= INT32_MAX / 8, "accumulate: too many tiles");
  const int64_t dh_ = h / d, dw_ = w / d;
  hipStream_t st = dh::as_stream(stream);
  if (dh_ == 0 || dw_ == 0) return DH_OK;
  DH_REQUIRE(dh_ * dw_ <= (int64_t)INT32_MAX, "accumulate: canvas too large");
  if (n > 0) {
    // The bin lists depend only on (origins, P, d, h, w): whole-slide prediction repeats the same
    // grid slide after slide, so the last plan (host CSR + its device copies) is kept and reused
    // when the origins compare equal -- no rebuild, no upload, no stream synchronisation.
    struct Plan {
      std::vector<int32_t> yx; int32_t P = 0, d = 0; int64_t h = 0, w = 0, nbins = 0, total = 0; int G = 0, bins_x = 0;
      int32_t *d_start = nullptr, *d_tiles = nullptr, *d_yx = nullptr; int device = -1;
    };
    static thread_local Plan plan;
    int dev_id = 0;
    DH_HIP(hipGetDevice(&dev_id));
    const bool hit = plan.device == dev_id && plan.P == P && plan.d == d && plan.h == h && plan.w == w &&
                     (int64_t)plan.yx.size() == 2 * n && memcmp(plan.yx.data(), yx_host, (size_t)n * 8) == 0;
    if (!hit) {
      const int G = std::max(1, std::min(P / d, 64));
      const int64_t bins_y = (dh_ + G - 1) / G, bins_x = (dw_ + G - 1) / G;
      const int64_t nbins = bins_y * bins_x;
      DH_REQUIRE(nbins < INT32_MAX, "accumulate: too many bins");
      std::vector<int32_t> start(nbins + 1, 0);
      auto span = [&](int64_t i, int64_t& cy0, int64_t& cy1, int64_t& cx0, int64_t& cx1) {
        const int64_t y = yx_host[2 * i], x = yx_host[2 * i + 1];
        cy0 = std::max<int64_t>(y / d, 0); cy1 = std::min<int64_t>((y + P) / d, dh_);
        cx0 = std::max<int64_t>(x / d, 0); cx1 = std::min<int64_t>((x + P) / d, dw_);
        return y >= 0 && x >= 0 && cy1 > cy0 && cx1 > cx0;
      };
      int64_t total = 0;
      for (int64_t i = 0; i < n; ++i) {
        int64_t a, b, c, e;
        DH_REQUIRE(yx_host[2 * i] >= 0 && yx_host[2 * i + 1] >= 0, "accumulate: negative origin");
        if (!span(i, a, b, c, e)) continue;
        for (int64_t by = a / G; by <= (b - 1) / G; ++by)
          for (int64_t bx = c / G; bx <= (e - 1) / G; ++bx) { ++start[by * bins_x + bx + 1]; ++total; }
      }
      DH_REQUIRE(total < INT32_MAX, "accumulate: incidence list too long");
      for (int64_t b = 0; b < nbins; ++b) start[b + 1] += start[b];
      std::vector<int32_t> fill(start.begin(), start.end() - 1), tiles((size_t)std::max<int64_t>(total, 1));
      for (int64_t i = 0; i < n; ++i) {
        int64_t a, b, c, e;
        if (!span(i, a, b, c, e)) continue;
        for (int64_t by = a / G; by <= (b - 1) / G; ++by)
          for (int64_t bx = c / G; bx <= (e - 1) / G; ++bx) tiles[fill[by * bins_x + bx]++] = (int32_t)i;
      }
      DH_HIP(hipStreamSynchronize(st));  // the previous plan's buffers may still be in use on this stream
      if (plan.d_start) { (void)hipFree(plan.d_start); (void)hipFree(plan.d_tiles); (void)hipFree(plan.d_yx); }
      plan = Plan();
      const size_t sb = (size_t)(nbins + 1) * 4, tb = (size_t)std::max<int64_t>(total, 1) * 4, yb = (size_t)n * 8;
      DH_HIP(hipMalloc((void**)&plan.d_start, sb));
      DH_HIP(hipMalloc((void**)&plan.d_tiles, tb));
      DH_HIP(hipMalloc((void**)&plan.d_yx, yb));
      DH_HIP(hipMemcpy(plan.d_start, start.data(), sb, hipMemcpyHostToDevice));
      DH_HIP(hipMemcpy(plan.d_tiles, tiles.data(), tb, hipMemcpyHostToDevice));
      DH_HIP(hipMemcpy(plan.d_yx, yx_host, yb, hipMemcpyHostToDevice));
      plan.yx.assign(yx_host, yx_host + 2 * n);
      plan.P = P; plan.d = d; plan.h = h; plan.w = w; plan.nbins = nbins; plan.total = total; plan.G = G;
      plan.bins_x = (int)bins_x; plan.device = dev_id;
    }
    if (plan.total > 0) {
      BinGeom g{plan.G, (int32_t)plan.bins_x, (int32_t)dh_, (int32_t)dw_, n_cls, P, d};
      hipLaunchKernelGGL(accumulate_bins_kernel, dim3((unsigned)plan.nbins), dim3(256), 0, st, logits, plan.d_yx,
                         plan.d_start, plan.d_tiles, g, canvas);
      DH_LAUNCH_CHECK();
    }
  }
  if (map) return dh_argmax_map(canvas, dh_ * dw_, n_cls, map, stream);
  return DH_OK;
}

// ---------------------------------------------------------------------------
// Visualisation of the class map (examples/predict_full_patched.py:81-113): HBM-bound byte work.
// ---------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void colorize_kernel(const int64_t* __restrict__ map, int64_t n, const uint8_t* __restrict__ lut,
                                                       int n_cls, uint8_t* __restrict__ rgb) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = map[i];
    uint8_t r = 0, g = 0, b = 0;
    if (c >= 0 && c < n_cls) { r = lut[3 * c]; g = lut[3 * c + 1]; b = lut[3 * c + 2]; }
    rgb[3 * i] = r; rgb[3 * i + 1] = g; rgb[3 * i + 2] = b;
  }
}
__global__ __launch_bounds__(256) void overlay_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ col, int64_t n,
                                                      double alpha, uint8_t* __restrict__ out) {
  const double beta = 1.0 - alpha;   // NumPy evaluates (1 - alpha) in float64 first
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (uint8_t)((double)img[i] * alpha + (double)col[i] * beta);   // values are in [0, 255]: the cast truncates
}
}  // namespace

extern "C" int dh_colorize_map(const int64_t* map, int64_t n, const uint8_t* lut, int32_t n_cls, uint8_t* rgb, void* stream) {
  DH_REQUIRE(n >= 0 && n_cls >= 0, "colorize: bad sizes");
  if (n == 0) return DH_OK;
  DH_REQUIRE(map && rgb && (lut || n_cls == 0), "colorize: null pointer");
  hipLaunchKernelGGL(colorize_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 65535)), dim3(256), 0, dh::as_stream(stream),
                     map, n, lut, n_cls, rgb);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_overlay_blend(const uint8_t* img, const uint8_t* col, int64_t n, double alpha, uint8_t* out, void* stream) {
  DH_REQUIRE(n >= 0 && alpha >= 0.0 && alpha <= 1.0, "overlay: bad arguments");
  if (n == 0) return DH_OK;
  DH_REQUIRE(img && col && out, "overlay: null pointer");
  hipLaunchKernelGGL(overlay_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 65535)), dim3(256), 0, dh::as_stream(stream),
                     img, col, n, alpha, out);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

// ---------------------------------------------------------------------------
// e1: the one exchange step of the sharded whole-slide prediction (every rank contributes the logits of its
// contiguous tile range, every rank receives all of them) as a C-ABI entry for hosts that are not torch.distributed programs.
// `comm` is the caller's ncclComm_t (RCCL).  The ncclAllGather that is called MUST come from the library instance that created `comm`
// (a second copy of RCCL handed a foreign communicator corrupts memory or hangs: ADVICE r4), so this shared object neither links
// against RCCL nor loads one on its own account.  Resolution order at the first exchange:
//   1. dh_set_rccl(handle): the host passes the dlopen handle of ITS RCCL (the exact answer; required when that copy was loaded
//      RTLD_LOCAL under a private path, e.g. a bundled torch/lib/librccl.so);
//   2. DH_RCCL_LIB=<file>: that file, which must ALREADY be loaded in the process (RTLD_NOLOAD probe; otherwise an error);
//   3. a global ncclAllGather symbol (RTLD_DEFAULT), then librccl.so.1 / librccl.so if already loaded (RTLD_NOLOAD).
// Nothing already loaded => DH_EINVAL: a process without RCCL cannot own a communicator.  (The Python shims exchange through
// torch.distributed, whose RCCL never gets here.)  rccl.h is included for ncclFloat32 / ncclResult_t only: no symbol is linked.
// ---------------------------------------------------------------------------
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <mutex>
namespace {
typedef ncclResult_t (*rccl_allgather_fn)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
typedef const char* (*rccl_errstr_fn)(ncclResult_t);
struct RcclEntry { rccl_allgather_fn allgather = nullptr; rccl_errstr_fn errstr = nullptr; std::string why; bool resolved = false; };
std::mutex g_rccl_mu;
RcclEntry g_rccl;
void rccl_from_handle(RcclEntry& r, void* h, const char* what) {
  void* sym = dlsym(h, "ncclAllGather");
  if (!sym) { r.why = std::string(what) + " has no ncclAllGather"; return; }
  r.allgather = reinterpret_cast<rccl_allgather_fn>(sym);
  r.errstr = reinterpret_cast<rccl_errstr_fn>(dlsym(h, "ncclGetErrorString"));
}
void rccl_resolve(RcclEntry& r) {
  r.resolved = true;
  if (const char* nm = getenv("DH_RCCL_LIB"); nm && *nm) {
    void* h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
    if (!h) { r.why = std::string("DH_RCCL_LIB=") + nm + " is not loaded in this process (the communicator's own library must be named)"; return; }
    rccl_from_handle(r, h, nm);
    return;
  }
  if (void* sym = dlsym(RTLD_DEFAULT, "ncclAllGather")) {
    r.allgather = reinterpret_cast<rccl_allgather_fn>(sym);
    r.errstr = reinterpret_cast<rccl_errstr_fn>(dlsym(RTLD_DEFAULT, "ncclGetErrorString"));
    return;
  }
  for (const char* nm : {"librccl.so.1", "librccl.so"})
    if (void* h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD)) { rccl_from_handle(r, h, nm); return; }
  r.why = "no RCCL is loaded in this process (pass the handle of the communicator's library with dh_set_rccl, or name it in DH_RCCL_LIB)";
}
}  // namespace

extern "C" int dh_set_rccl(void* dl_handle) {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  g_rccl = RcclEntry();
  if (!dl_handle) return DH_OK;   // back to automatic resolution at the next exchange
  g_rccl.resolved = true;
  rccl_from_handle(g_rccl, dl_handle, "the handle given to dh_set_rccl");
  DH_REQUIRE(g_rccl.allgather, "dh_set_rccl: %s", g_rccl.why.c_str());
  return DH_OK;
}

extern "C" int dh_allgather_logits(void* comm, const float* send_dev, float* recv_dev, int64_t n_per_rank, int32_t n_cls, void* stream) {
  DH_REQUIRE(comm, "allgather_logits: null communicator");
  DH_REQUIRE(n_per_rank >= 0 && n_cls > 0, "allgather_logits: bad sizes");
  if (n_per_rank == 0) return DH_OK;
  DH_REQUIRE(send_dev && recv_dev, "allgather_logits: null pointer");
  RcclEntry e;
  {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (!g_rccl.resolved) rccl_resolve(g_rccl);
    e = g_rccl;
  }
  DH_REQUIRE(e.allgather, "allgather_logits: %s", e.why.empty() ? "RCCL not found" : e.why.c_str());
  const ncclResult_t rc = e.allgather(send_dev, recv_dev, (size_t)n_per_rank * (size_t)n_cls, ncclFloat32, static_cast<ncclComm_t>(comm), dh::as_stream(stream));
  if (rc != ncclSuccess) {
    dh::set_error("allgather_logits: ncclAllGather failed: %s (%d)", e.errstr ? e.errstr(rc) : "?", (int)rc);
    return DH_EHIP;
  }
  return DH_OK;
}
