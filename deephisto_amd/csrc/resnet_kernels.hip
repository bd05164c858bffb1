// ResNet-18 patch-classifier forward for MI355X (gfx950), hand-written HIP.
//
// Replaces `model(features)` of the network built by
// models/patch_cls_simple/model.py:5-11 (torchvision resnet18 + fc[n_cls,512]) as
// called from examples/predict_full_patched.py:77 (eval mode).
//
// Design:
//  * activations live in HBM pixel-major: NHWC for float32, channel-blocked [image][C/32][H][W][32] for bf16
//    inference, so the reduction index of every convolution, (tap, cin), is contiguous in cin: a 64-byte channel
//    chunk of one pixel is one LDS row, and a lane's MFMA B-fragment is one 16-byte read;
//  * each conv is an implicit GEMM  D[cout][pixel] = sum_k W[cout][k] * X[k][pixel]
//    on v_mfma_f32_32x32x16_bf16 (bf16 mode) or v_mfma_f32_32x32x2_f32 (f32 mode, exact f32 products, used for
//    the 1e-4 parity runs).  cout is the MFMA row index so every lane ends up with 4 consecutive couts of one
//    pixel -> 8/16-byte stores with BN scale/shift, residual add and ReLU fused in the epilogue;
//  * 3x3 convs (stride 1 and 2; the 1x1 stride-2 downsample rides on the stride-2 kernel): conv3x3.inc --
//    persistent 8-wave workgroups, one per CU, 64 couts x 128..512 pixels per tile, LDS-DMA ring of
//    [weight slab | input window] stages, all 9 taps read shifted views of the staged window;
//  * bf16 stem: stem_pool.inc (conv 7x7/2 + BN + ReLU + maxpool fused, persistent); the generic conv_kernel
//    below serves the f32 1x1 convs of training and the debug hook;
//  * weights are pre-packed on the host in MFMA fragment order, so staging them is a linear copy and reading them
//    is conflict-free `base + lane*16`;
//  * host-side caches (lookup tables, zero page, function attributes) are keyed by HIP device and guarded by one
//    mutex: one process may drive several devices / call from several threads (one handle per thread).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "dh_common.h"
#include "conv3_tables_host.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int CHUNK_BYTES = 64;   // channel bytes of one pixel staged per pass
constexpr int PIX_PITCH = 80;     // LDS bytes per staged pixel (64 + 16 pad: bank spread)
constexpr int FRAG_BYTES = 1024;  // one MFMA operand fragment: 64 lanes x 16 B
constexpr int SLAB_TAP = 4 * FRAG_BYTES;  // per tap: 2 k-steps x 2 cout-tiles

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> { static constexpr int ESZ = 4; };
template <> struct ElemTraits<__bf16> { static constexpr int ESZ = 2; };

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float div255f(uint32_t k) {
  const float r = 1.0f / 255.0f;
  const float kf = (float)k;
  const float q = kf * r;
  return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, kf), r, q);
}

template <typename T>
__device__ __forceinline__ void mma_frag(f32x16& acc, const uint4& a, const uint4& b) {
  if constexpr (sizeof(T) == 2) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                  __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
}

// Epilogue shared by stem and conv kernels.  acc[mt][nt] holds D[cout][pixel]:
// lane: pixel = lane&31 of n-tile nt; register r: cout = 32*mt + (r&3) + 8*(r>>2) + 4*(lane>>5).
// y = acc*scale + shift (+ residual) (ReLU) -> 4 consecutive couts per store.
template <typename T>
__device__ __forceinline__ void store_group(const f32x16& acc, int g, const float* __restrict__ scale,
                                            const float* __restrict__ shift, const T* __restrict__ res,
                                            T* __restrict__ out, int64_t off, int co, bool relu) {
  const float4 sc = *reinterpret_cast<const float4*>(scale + co);
  const float4 sh = *reinterpret_cast<const float4*>(shift + co);
  float v[4] = {__builtin_fmaf(acc[4 * g + 0], sc.x, sh.x), __builtin_fmaf(acc[4 * g + 1], sc.y, sh.y),
                __builtin_fmaf(acc[4 * g + 2], sc.z, sh.z), __builtin_fmaf(acc[4 * g + 3], sc.w, sh.w)};
  if (res) {
    if constexpr (sizeof(T) == 2) {
      const uint2 r = *reinterpret_cast<const uint2*>(res + off);
      v[0] += bf16_bits_to_f32(r.x & 0xFFFFu); v[1] += bf16_bits_to_f32(r.x >> 16);
      v[2] += bf16_bits_to_f32(r.y & 0xFFFFu); v[3] += bf16_bits_to_f32(r.y >> 16);
    } else {
      const float4 r = *reinterpret_cast<const float4*>(res + off);
      v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
    }
  }
  if (relu) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
  }
  if constexpr (sizeof(T) == 2) {
    uint2 w;
    w.x = f32_to_bf16(v[0]) | (f32_to_bf16(v[1]) << 16);
    w.y = f32_to_bf16(v[2]) | (f32_to_bf16(v[3]) << 16);
    *reinterpret_cast<uint2*>(out + off) = w;
  } else {
    *reinterpret_cast<float4*>(out + off) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// ---------------------------------------------------------------------------
// Generic conv (KS x KS, stride STRIDE, pad KS/2), NHWC, Cin % (64/ESZ) == 0,
// Cout % 64 == 0.  HALO (3x3, stride 1 only): stage patch+halo once per chunk.
// ---------------------------------------------------------------------------
struct ConvParams {
  const void* in; const void* w; const float* scale; const float* shift; const void* res; void* out;
  int B, Hi, Wi, Cin, Ho, Wo, Cout;
  int TH, TW, IMGS;       // output patch per workgroup: IMGS images x TH x TW = 256 pixels
  int tiles_y, tiles_x;   // patches per image
  int relu;
};

constexpr int MAX_HALO_PIECES = 7;  // ceil(4 * IMGS*(TH+2)*(TW+2) / 256), worst case 4*400/256

template <typename T, int KS, int STRIDE, bool HALO>
__global__ __launch_bounds__(256, 2) void conv_kernel(const ConvParams p) {
  static_assert(!HALO || (KS == 3 && STRIDE == 1), "halo staging is for 3x3 stride 1");
  constexpr int ESZ = ElemTraits<T>::ESZ;
  constexpr int CPC = CHUNK_BYTES / ESZ;  // channels per chunk
  constexpr int TAPS = KS * KS, PAD = KS / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* w_lds = smem;
  char* a_lds = smem + TAPS * SLAB_TAP;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ncb = p.Cout >> 6;
  const int cb = blockIdx.x % ncb, pt = blockIdx.x / ncb;
  const int tiles_per_img = p.tiles_y * p.tiles_x;
  const int img0 = (pt / tiles_per_img) * p.IMGS;
  const int tile = pt % tiles_per_img;
  const int oy0 = (tile / p.tiles_x) * p.TH, ox0 = (tile % p.tiles_x) * p.TW;
  const int nchunks = p.Cin / CPC;
  const int HW2 = p.TW + 2, HH2 = p.TH + 2;
  const char* in = static_cast<const char*>(p.in);

  // this lane's two pixels (one per n-tile)
  int pimg[2], poy[2], pox[2], b_off[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int pidx = wave * 64 + nt * 32 + (lane & 31);
    const int img = pidx / (p.TH * p.TW), rem = pidx % (p.TH * p.TW);
    const int ty = rem / p.TW, tx = rem % p.TW;
    pimg[nt] = img0 + img; poy[nt] = oy0 + ty; pox[nt] = ox0 + tx;
    b_off[nt] = (HALO ? ((img * HH2 + ty) * HW2 + tx) : pidx) * PIX_PITCH + 16 * (lane >> 5);
  }

  // staging descriptors (global byte offset at chunk 0, LDS byte offset); -1 = zero fill
  uint32_t g_off[MAX_HALO_PIECES];
  int l_off[MAX_HALO_PIECES];
  bool g_ok[MAX_HALO_PIECES];
  const int n_pieces = HALO ? p.IMGS * HH2 * HW2 * 4 : 0;
  if constexpr (HALO) {
#pragma unroll
    for (int j = 0; j < MAX_HALO_PIECES; ++j) {
      const int i = tid + j * 256;
      const int hp = i >> 2, q = i & 3;
      const int img = hp / (HH2 * HW2), r = hp % (HH2 * HW2);
      const int iy = oy0 + r / HW2 - 1, ix = ox0 + r % HW2 - 1, b = img0 + img;
      g_ok[j] = i < n_pieces && b < p.B && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
      g_off[j] = (uint32_t)((((int64_t)b * p.Hi + iy) * p.Wi + ix) * p.Cin * ESZ + q * 16);
      l_off[j] = i < n_pieces ? hp * PIX_PITCH + q * 16 : -1;
    }
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const char* wsrc = static_cast<const char*>(p.w) + (int64_t)cb * nchunks * TAPS * SLAB_TAP;

  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();  // previous chunk's LDS reads are done
    // weights: linear copy of the fragment-ordered slab
    for (int i = tid * 16; i < TAPS * SLAB_TAP; i += 256 * 16)
      *reinterpret_cast<uint4*>(w_lds + i) =
          *reinterpret_cast<const uint4*>(wsrc + (int64_t)ch * TAPS * SLAB_TAP + i);
    if constexpr (HALO) {
#pragma unroll
      for (int j = 0; j < MAX_HALO_PIECES; ++j) {
        if (l_off[j] >= 0) {
          uint4 v = make_uint4(0, 0, 0, 0);
          if (g_ok[j]) v = *reinterpret_cast<const uint4*>(in + g_off[j] + ch * CHUNK_BYTES);
          *reinterpret_cast<uint4*>(a_lds + l_off[j]) = v;
        }
      }
    }
#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
      const int kh = tap / KS, kw = tap % KS;
      if constexpr (!HALO) {
        if (tap > 0) __syncthreads();  // previous tap's reads done before overwrite
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = tid + j * 256;
          const int pidx = i >> 2, q = i & 3;
          const int img = pidx / (p.TH * p.TW), rem = pidx % (p.TH * p.TW);
          const int oy = oy0 + rem / p.TW, ox = ox0 + rem % p.TW, b = img0 + img;
          const int iy = oy * STRIDE + kh - PAD, ix = ox * STRIDE + kw - PAD;
          uint4 v = make_uint4(0, 0, 0, 0);
          if (b < p.B && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi)
            v = *reinterpret_cast<const uint4*>(
                in + (((int64_t)b * p.Hi + iy) * p.Wi + ix) * p.Cin * ESZ + ch * CHUNK_BYTES + q * 16);
          *reinterpret_cast<uint4*>(a_lds + pidx * PIX_PITCH + q * 16) = v;
        }
      }
      if (!HALO || tap == 0) __syncthreads();
      const int tap_off = HALO ? (kh * HW2 + kw) * PIX_PITCH : 0;
      const char* wt = w_lds + tap * SLAB_TAP + lane * 16;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const uint4 a0 = *reinterpret_cast<const uint4*>(wt + (ks * 2 + 0) * FRAG_BYTES);
        const uint4 a1 = *reinterpret_cast<const uint4*>(wt + (ks * 2 + 1) * FRAG_BYTES);
        const uint4 b0 = *reinterpret_cast<const uint4*>(a_lds + b_off[0] + tap_off + ks * 32);
        const uint4 b1 = *reinterpret_cast<const uint4*>(a_lds + b_off[1] + tap_off + ks * 32);
        mma_frag<T>(acc[0][0], a0, b0);
        mma_frag<T>(acc[0][1], a0, b1);
        mma_frag<T>(acc[1][0], a1, b0);
        mma_frag<T>(acc[1][1], a1, b1);
      }
    }
  }

  // epilogue
  const T* res = static_cast<const T*>(p.res);
  T* out = static_cast<T*>(p.out);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    if (pimg[nt] < p.B && poy[nt] < p.Ho && pox[nt] < p.Wo) {
      const int64_t pix = ((int64_t)pimg[nt] * p.Ho + poy[nt]) * p.Wo + pox[nt];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = cb * 64 + mt * 32 + g * 8 + 4 * (lane >> 5);
          store_group<T>(acc[mt][nt], g, p.scale, p.shift, res, out, pix * p.Cout + co, co, p.relu != 0);
        }
    }
  }
}

#include "conv3x3.inc"
#include "stem_pool.inc"

// ---------------------------------------------------------------------------
// Stem: conv 7x7 / stride 2 / pad 3, 3 -> 64, + BN + ReLU, output NHWC.
// Source is either the model input float32 NCHW [n,3,P,P] (SRC_NCHW) or the uint8
// HWC slide plus tile origins (SRC_U8: gather and /255 fused, a2+a5+a6).
// Workgroup = 64 couts x (8 x 32 output pixels).  The 21 x 69 input window is
// staged in LDS in HWC element order, normalised; a tap row kh contributes a
// 32-wide (bf16) / 22-wide (f32) k-slice: slot 0 and the slots past 21 carry zero
// weights, which keeps every fragment read 4-byte aligned (element 6*tx + i).
// ---------------------------------------------------------------------------
constexpr int STEM_TH = 8, STEM_TW = 32;
constexpr int STEM_ROWS = 2 * STEM_TH + 5;   // 21 input rows
constexpr int STEM_ROWE = 224;               // elements per staged row (>= 6*31 + 32)
constexpr int STEM_REAL = 1 + 3 * (2 * STEM_TW + 5);  // 208 meaningful elements (slot 0 = dummy)

struct StemParams {
  const float* x_nchw; const uint8_t* slide; const int32_t* yx; int64_t row_bytes;
  const void* w; const float* scale; const float* shift; void* out;
  int B, P, Ho, Wo, tiles_y, tiles_x;
  int relu;
};

template <typename T, bool SRC_U8>
__global__ __launch_bounds__(256, 2) void stem_kernel(const StemParams p) {
  constexpr int ESZ = ElemTraits<T>::ESZ;
  constexpr int W_BYTES = (ESZ == 2) ? 7 * 2 * 2 * FRAG_BYTES : 7 * 11 * 2 * 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* w_lds = smem;
  T* e_lds = reinterpret_cast<T*>(smem + W_BYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_per_img = p.tiles_y * p.tiles_x;
  const int b = blockIdx.x / tiles_per_img, tile = blockIdx.x % tiles_per_img;
  const int oy0 = (tile / p.tiles_x) * STEM_TH, ox0 = (tile % p.tiles_x) * STEM_TW;
  const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;

  for (int i = tid * 16; i < W_BYTES; i += 256 * 16)
    *reinterpret_cast<uint4*>(w_lds + i) = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.w) + i);

  int ty0 = 0, tx0 = 0;
  if constexpr (SRC_U8) { ty0 = p.yx[2 * b]; tx0 = p.yx[2 * b + 1]; }
  // window staging in two phases: every load of the thread's 19 elements is issued before the first LDS store (a load + store per
  // iteration of a rolled loop serialised on the load latency: 18 round trips per tile, ~90 % of the bf16 kernel's time in training)
  constexpr int NE = STEM_ROWS * STEM_ROWE;
  auto fetch = [&](int i) __attribute__((always_inline)) -> float {   // element i of the window image (raw value; 0 outside the image)
    const int r = i / STEM_ROWE, e = i % STEM_ROWE;
    const int q = e - 1, ix = ix0 + q / 3, c = q % 3, iy = iy0 + r;
    float x = 0.f;
    if (i < NE && e >= 1 && e < STEM_REAL && iy >= 0 && iy < p.P && ix >= 0 && ix < p.P) {
      if constexpr (SRC_U8) x = (float)p.slide[(int64_t)(ty0 + iy) * p.row_bytes + (int64_t)(tx0 + ix) * 3 + c];
      else x = p.x_nchw[(((int64_t)b * 3 + c) * p.P + iy) * p.P + ix];
    }
    return x;
  };
  if constexpr (ESZ == 2) {   // element pairs: one 4-byte LDS store per two elements
    constexpr int NP = NE / 2, NIT = (NP + 255) / 256;
    float v0[NIT], v1[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) { const int j = tid + 256 * k; v0[k] = fetch(j < NP ? 2 * j : NE); v1[k] = fetch(j < NP ? 2 * j + 1 : NE); }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int j = tid + 256 * k;
      const float x0 = SRC_U8 ? div255f((uint32_t)v0[k]) : v0[k], x1 = SRC_U8 ? div255f((uint32_t)v1[k]) : v1[k];
      if (j < NP) reinterpret_cast<uint32_t*>(e_lds)[j] = f32_to_bf16(x0) | (f32_to_bf16(x1) << 16);
    }
  } else {
    constexpr int NIT = (NE + 255) / 256;
    float v[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) v[k] = fetch(tid + 256 * k);
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = tid + 256 * k;
      if (i < NE) reinterpret_cast<float*>(e_lds)[i] = SRC_U8 ? div255f((uint32_t)v[k]) : v[k];
    }
  }
  __syncthreads();

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int tx = lane & 31, h = lane >> 5;
  // n-tile nt = output row ty = 2*wave + nt; input row for tap row kh: 2*ty + kh
#pragma unroll 1
  for (int kh = 0; kh < 7; ++kh) {
    if constexpr (ESZ == 2) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const char* wt = w_lds + ((kh * 2 + t) * 2) * FRAG_BYTES + lane * 16;
        const uint4 a0 = *reinterpret_cast<const uint4*>(wt);
        const uint4 a1 = *reinterpret_cast<const uint4*>(wt + FRAG_BYTES);
        uint4 bb[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int row = 2 * (2 * wave + nt) + kh;
          const uint32_t* src = reinterpret_cast<const uint32_t*>(
              reinterpret_cast<const char*>(e_lds) + (row * STEM_ROWE + 6 * tx + 16 * t + 8 * h) * 2);
          bb[nt] = make_uint4(src[0], src[1], src[2], src[3]);
        }
        mma_frag<T>(acc[0][0], a0, bb[0]);
        mma_frag<T>(acc[0][1], a0, bb[1]);
        mma_frag<T>(acc[1][0], a1, bb[0]);
        mma_frag<T>(acc[1][1], a1, bb[1]);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 11; ++s) {
        const float* wt = reinterpret_cast<const float*>(w_lds) + ((kh * 11 + s) * 2) * 64 + lane;
        const float a0 = wt[0], a1 = wt[64];
        float bv[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int row = 2 * (2 * wave + nt) + kh;
          bv[nt] = reinterpret_cast<const float*>(e_lds)[row * STEM_ROWE + 6 * tx + 2 * s + h];
        }
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[1], acc[1][1], 0, 0, 0);
      }
    }
  }

  T* out = static_cast<T*>(p.out);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int oy = oy0 + 2 * wave + nt, ox = ox0 + tx;
    if (oy < p.Ho && ox < p.Wo) {
      const int64_t pix = ((int64_t)b * p.Ho + oy) * p.Wo + ox;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = mt * 32 + g * 8 + 4 * h;
          store_group<T>(acc[mt][nt], g, p.scale, p.shift, (const T*)nullptr, out, pix * 64 + co, co, p.relu != 0);
        }
    }
  }
}

// ---------------------------------------------------------------------------
// maxpool 3x3 / stride 2 / pad 1, NHWC; one thread = 16 bytes of channels.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                      int B, int Hi, int Wi, int C, int Ho, int Wo) {
  constexpr int EPV = 16 / (int)sizeof(T);
  const int cv = C / EPV;
  const int64_t total = (int64_t)B * Ho * Wo * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv);
    int64_t r = i / cv;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float m[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) m[e] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = 2 * oy + dy - 1;
      if (iy < 0 || iy >= Hi) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = 2 * ox + dx - 1;
        if (ix < 0 || ix >= Wi) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(in + (((int64_t)b * Hi + iy) * Wi + ix) * C + c * EPV);
        const uint32_t u[4] = {v.x, v.y, v.z, v.w};
        if constexpr (sizeof(T) == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            m[2 * e] = fmaxf(m[2 * e], bf16_bits_to_f32(u[e] & 0xFFFFu));
            m[2 * e + 1] = fmaxf(m[2 * e + 1], bf16_bits_to_f32(u[e] >> 16));
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], __uint_as_float(u[e]));
        }
      }
    }
    uint4 o;
    if constexpr (sizeof(T) == 2) {
      o.x = f32_to_bf16(m[0]) | (f32_to_bf16(m[1]) << 16); o.y = f32_to_bf16(m[2]) | (f32_to_bf16(m[3]) << 16);
      o.z = f32_to_bf16(m[4]) | (f32_to_bf16(m[5]) << 16); o.w = f32_to_bf16(m[6]) | (f32_to_bf16(m[7]) << 16);
    } else {
      o = make_uint4(__float_as_uint(m[0]), __float_as_uint(m[1]), __float_as_uint(m[2]), __float_as_uint(m[3]));
    }
    *reinterpret_cast<uint4*>(out + i * EPV) = o;
  }
}

// ---------------------------------------------------------------------------
// global average pool + fc: one workgroup per image; f32 arithmetic.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fc_kernel(const T* __restrict__ in, int HW, int C, int blocked,
                                                         const float* __restrict__ fc_w,
                                                         const float* __restrict__ fc_b, int n_cls,
                                                         float* __restrict__ logits) {
  // one workgroup per image; lane = 16 bytes of channels, waves split the pixels (C == 512:
  // bf16 -> 64 lanes x 8 channels per pixel; f32 -> 128 lanes x 4 channels, 2 pixel groups)
  constexpr int EPV = 16 / (int)sizeof(T);
  __shared__ float part[4][512];
  __shared__ float pooled[512];
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int lanes_per_px = C / EPV;               // 64 or 128
  const int groups = 256 / lanes_per_px;          // 4 or 2 pixel groups
  const int grp = tid / lanes_per_px, cl = tid % lanes_per_px;
  float s[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) s[e] = 0.f;
  for (int q = grp; q < HW; q += groups) {
    // NHWC, or channel-blocked [image][C/32][HW][32] (a lane's 16 bytes never straddle a 32-channel chunk)
    const int64_t off = blocked ? (int64_t)b * HW * C + (int64_t)((cl * EPV) >> 5) * HW * 32 + q * 32 + ((cl * EPV) & 31)
                                : ((int64_t)b * HW + q) * C + cl * EPV;
    const uint4 v = *reinterpret_cast<const uint4*>(in + off);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[2 * e] += bf16_bits_to_f32(u[e] & 0xFFFFu); s[2 * e + 1] += bf16_bits_to_f32(u[e] >> 16); }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += __uint_as_float(u[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < EPV; ++e) part[grp][cl * EPV + e] = s[e];
  __syncthreads();
  const float inv = 1.0f / (float)HW;
  for (int c = tid; c < C; c += 256) {
    float t = 0.f;
    for (int g = 0; g < groups; ++g) t += part[g][c];
    pooled[c] = t * inv;
  }
  __syncthreads();
  for (int k = 0; k < n_cls; ++k) {
    float t = 0.f;
    for (int c = tid; c < C; c += 256) t = __builtin_fmaf(pooled[c], fc_w[k * C + c], t);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = t;
    __syncthreads();
    if (tid == 0) logits[(int64_t)b * n_cls + k] = red[0] + red[1] + red[2] + red[3] + fc_b[k];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct ConvLayer {
  std::string name;  // state_dict prefix of the conv ("layer1.0.conv1"); BN is its sibling
  std::string bn;
  int cin, cout, ks, stride;
  void* w_dev = nullptr;        // packed weights
  void* w2_dev = nullptr;       // bf16 inference, stride-2 3x3 convs with cout % 128 == 0 and their downsample convs: the wide (HALF) packing
  float* scale_dev = nullptr;   // [cout]
  float* shift_dev = nullptr;
};

inline uint16_t host_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x7FFFFFu)) return (uint16_t)((u >> 16) | 0x40);
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

}  // namespace

struct dh_train;
struct dh_resnet18 {
  dh_train* train = nullptr;  // training state (train.inc), owned
  int n_classes = 0;
  int dtype = DH_DTYPE_F32;
  std::map<std::string, std::vector<float>> params;  // host copies by state_dict name
  std::vector<ConvLayer> convs;                      // index 0 = stem
  float* fc_w_dev = nullptr;
  float* fc_b_dev = nullptr;
  bool finalized = false;
  // activation workspace
  void* ws = nullptr;
  size_t ws_bytes = 0;
  int esz() const { return dtype == DH_DTYPE_F32 ? 4 : 2; }
};

namespace {

void build_topology(dh_resnet18* net) {
  net->convs.clear();
  net->convs.push_back({"conv1", "bn1", 3, 64, 7, 2});
  const int ch[4] = {64, 128, 256, 512};
  int cin = 64;
  for (int s = 0; s < 4; ++s) {
    for (int blk = 0; blk < 2; ++blk) {
      const std::string pre = "layer" + std::to_string(s + 1) + "." + std::to_string(blk);
      const int stride = (blk == 0 && s > 0) ? 2 : 1;
      const int bcin = blk == 0 ? cin : ch[s];
      net->convs.push_back({pre + ".conv1", pre + ".bn1", bcin, ch[s], 3, stride});
      net->convs.push_back({pre + ".conv2", pre + ".bn2", ch[s], ch[s], 3, 1});
      if (blk == 0 && s > 0)
        net->convs.push_back({pre + ".downsample.0", pre + ".downsample.1", bcin, ch[s], 1, 2});
    }
    cin = ch[s];
  }
}

int64_t expected_elems(const dh_resnet18* net, const std::string& name) {
  if (name == "fc.weight") return (int64_t)net->n_classes * 512;
  if (name == "fc.bias") return net->n_classes;
  for (const auto& c : net->convs) {
    if (name == c.name + ".weight") return (int64_t)c.cout * c.cin * c.ks * c.ks;
    for (const char* s : {".weight", ".bias", ".running_mean", ".running_var"})
      if (name == c.bn + s) return c.cout;
    if (name == c.bn + ".num_batches_tracked") return 1;
  }
  return -1;
}

// Pack [cout][cin][ks][ks] f32 into fragment order:
//   [cb = cout/64][chunk][tap][ks2][mt][lane][16 B];  element e of lane l:
//   co = 64cb + 32mt + (l&31); ci = chunk*CPC + ks2*(CPC/2) + (l>>5)*EPL + e.
void pack_conv_weights(const float* w, int cout, int cin, int ks, int esz, std::vector<uint8_t>& out) {
  const int cpc = CHUNK_BYTES / esz, epl = 16 / esz, taps = ks * ks;
  const int ncb = cout / 64, nch = cin / cpc;
  out.assign((size_t)ncb * nch * taps * SLAB_TAP, 0);
  size_t o = 0;
  for (int cb = 0; cb < ncb; ++cb)
    for (int ch = 0; ch < nch; ++ch)
      for (int tap = 0; tap < taps; ++tap)
        for (int k2 = 0; k2 < 2; ++k2)
          for (int mt = 0; mt < 2; ++mt)
            for (int l = 0; l < 64; ++l)
              for (int e = 0; e < epl; ++e) {
                const int co = cb * 64 + mt * 32 + (l & 31);
                const int ci = ch * cpc + k2 * (cpc / 2) + (l >> 5) * epl + e;
                const float v = w[((size_t)co * cin + ci) * taps + tap];
                if (esz == 4) { memcpy(&out[o], &v, 4); o += 4; }
                else { const uint16_t hb = host_bf16(v); memcpy(&out[o], &hb, 2); o += 2; }
              }
}

// The wide stride-2 variant's packing (conv3x3.inc, HALF): [cb = cout/128][hc = cin/16][tap][mt (4)][lane][8 bf16]; element e of lane l:
//   co = 128 cb + 32 mt + (l & 31); ci = 16 hc + 8 (l >> 5) + e.   (ks = 1: one "tap" per (cb, hc): 4 KiB)
void pack_conv_weights_wide(const float* w, int cout, int cin, int ks, std::vector<uint8_t>& out) {
  const int taps = ks * ks, ncb = cout / 128, nhc = cin / 16;
  out.assign((size_t)ncb * nhc * taps * SLAB_TAP, 0);
  size_t o = 0;
  for (int cb = 0; cb < ncb; ++cb)
    for (int hc = 0; hc < nhc; ++hc)
      for (int tap = 0; tap < taps; ++tap)
        for (int mt = 0; mt < 4; ++mt)
          for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 8; ++e) {
              const int co = cb * 128 + mt * 32 + (l & 31), ci = hc * 16 + (l >> 5) * 8 + e;
              const uint16_t hb = host_bf16(w[((size_t)co * cin + ci) * taps + tap]);
              memcpy(&out[o], &hb, 2); o += 2;
            }
}

// Stem weights [64][3][7][7].  bf16: [kh][t][mt][lane][8], slot i = 16t + 8h + j,
// value W[co][c][kh][kw] for i-1 = 3kw + c in [0,21), else 0.
// f32: [kh][s][mt][lane], slot i = 2s + h.
void pack_stem_weights(const float* w, int esz, std::vector<uint8_t>& out) {
  auto slot = [&](int co, int kh, int i) -> float {
    if (i < 1 || i > 21) return 0.f;
    const int kw = (i - 1) / 3, c = (i - 1) % 3;
    return w[((size_t)(co * 3 + c) * 7 + kh) * 7 + kw];
  };
  if (esz == 2) {
    out.assign((size_t)7 * 2 * 2 * FRAG_BYTES, 0);
    size_t o = 0;
    for (int kh = 0; kh < 7; ++kh)
      for (int t = 0; t < 2; ++t)
        for (int mt = 0; mt < 2; ++mt)
          for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
              const uint16_t hb = host_bf16(slot(mt * 32 + (l & 31), kh, 16 * t + 8 * (l >> 5) + j));
              memcpy(&out[o], &hb, 2); o += 2;
            }
  } else {
    out.assign((size_t)7 * 11 * 2 * 256, 0);
    size_t o = 0;
    for (int kh = 0; kh < 7; ++kh)
      for (int s = 0; s < 11; ++s)
        for (int mt = 0; mt < 2; ++mt)
          for (int l = 0; l < 64; ++l) {
            const float v = slot(mt * 32 + (l & 31), kh, 2 * s + (l >> 5));
            memcpy(&out[o], &v, 4); o += 4;
          }
  }
}

// Stem weights for the fused bf16 stem (stem_pool.inc): [k-step (11)][cout half][lane][8], two kernel rows per three k-steps
// (sp_kslot maps a k-slot to its kernel row and 24-slot row image; slot s carries W[co][c][kh][kw] for s-1 = 3kw + c in [0,21)).
void pack_stem_pool_weights(const float* w, std::vector<uint8_t>& out) {
  out.assign((size_t)SP_WBYTES, 0);
  size_t o = 0;
  for (int ks = 0; ks < SP_KS; ++ks)
    for (int mt = 0; mt < 2; ++mt)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          int kh, sl;
          sp_kslot(ks, l >> 5, j, &kh, &sl);
          float v = 0.f;
          if (kh >= 0 && sl >= 1 && sl <= 21) {
            const int co = mt * 32 + (l & 31), kw = (sl - 1) / 3, c = (sl - 1) % 3;
            v = w[((size_t)(co * 3 + c) * 7 + kh) * 7 + kw];
          }
          const uint16_t hb = host_bf16(v);
          memcpy(&out[o], &hb, 2); o += 2;
        }
}

// Optional in-library timing of the dominant kernel (3x3 stride-1 conv): HIP events on
// the launch stream around sampled launches, summed by dh_profile_stop (bench.py's
// `roofline.achieved`).  Off by default; costs nothing when off.
bool g_stamps_on = false;
unsigned long long* g_stamps_dev = nullptr;

struct Profiler {
  bool on = false;
  int every = 1, counter = 0;
  std::vector<hipEvent_t> ev;  // pairs
  size_t used = 0;
  double flops = 0.0;
} g_prof;

// ---- per-device host state ---------------------------------------------------------------------------------
std::mutex g_host_mu;   // guards every cache below (ctypes releases the GIL: two threads may be in here)
inline int current_device() { int d = 0; (void)hipGetDevice(&d); return d; }

// hipFuncSetAttribute is per device: remember which (kernel, device) pairs are done
int ensure_dyn_lds(const void* fn, int bytes) {
  static std::set<std::pair<const void*, int>> done;
  const std::pair<const void*, int> key(fn, current_device());
  std::lock_guard<std::mutex> lk(g_host_mu);
  if (done.count(key)) return DH_OK;
  DH_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.insert(key);
  return DH_OK;
}

template <typename T, int KS, int STRIDE, bool HALO>
int launch_conv(const ConvParams& p, hipStream_t st) {
  const int npx_lds = HALO ? p.IMGS * (p.TH + 2) * (p.TW + 2) : 256;
  const size_t lds = (size_t)KS * KS * SLAB_TAP + (size_t)npx_lds * PIX_PITCH;
  const int blocks_per_img = p.tiles_y * p.tiles_x;
  const int groups = ((p.B + p.IMGS - 1) / p.IMGS) * blocks_per_img;
  const int grid = groups * (p.Cout / 64);
  if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(&conv_kernel<T, KS, STRIDE, HALO>), 96 * 1024)) return rc;
  hipLaunchKernelGGL((conv_kernel<T, KS, STRIDE, HALO>), dim3(grid), dim3(256), lds, st, p);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

std::map<int, void*> g_zero_page;  // per device: 1 KiB of zeros, the DMA source of padding pixels
int zero_page(const void** out) {
  const int dev = current_device();
  std::lock_guard<std::mutex> lk(g_host_mu);
  auto it = g_zero_page.find(dev);
  if (it == g_zero_page.end()) {
    void* z = nullptr;
    DH_HIP(hipMalloc(&z, 1024));
    DH_HIP(hipMemset(z, 0, 1024));
    it = g_zero_page.emplace(dev, z).first;
  }
  *out = it->second;
  return DH_OK;
}

// Host-built lookup tables of the conv3x3 kernel, cached per (device, layer shape): per-thread geometry and per-tile
// decode, a few KiB each.  The key holds the batch only through the tile count; the cache is bounded (a caller that
// sweeps batch sizes would otherwise grow it without end): past CONV3_TABLE_CAP entries of a device it is emptied
// (hipFree waits for kernels that still read a table).
struct Conv3Tables { int* lane = nullptr; int4* tile = nullptr; unsigned* mask = nullptr; };
std::map<std::vector<int>, Conv3Tables> g_conv3_tables;
constexpr size_t CONV3_TABLE_CAP = 256;

template <int STRIDE, int NT, int WAVES, int ESZ, int MT, bool HALF = false>
int conv3_tables(const Conv3Params& p, int ncb, int groups_img, Conv3Tables* out, int grid_override = 0) {
  const std::vector<int> key = {current_device(), STRIDE, NT, WAVES, MT + (HALF ? 100 : 0), ESZ, p.TH, p.TW, p.IMGS, p.HR, p.HC, p.HP, p.HPH, p.Hi, p.Wi,
                                p.Cin, p.B, p.tiles_y, p.tiles_x, ncb, p.n_win_instr, p.WTAIL, p.FIT, p.IP, p.in_px_bytes, p.Ho, p.Wo, p.Cout, p.out_px,
                                p.out_cb, (int)(p.o_img & 0x7FFFFFFF), (int)(p.o_img >> 31), p.o_row, p.o_px, p.o_base, grid_override, p.iters,
                                p.r_row, p.r_px, p.r_cb, p.r_base};
  std::lock_guard<std::mutex> lk(g_host_mu);
  auto it = g_conv3_tables.find(key);
  if (it != g_conv3_tables.end()) { *out = it->second; return DH_OK; }
  if (g_conv3_tables.size() >= CONV3_TABLE_CAP) {
    for (auto& kv : g_conv3_tables) { (void)hipFree(kv.second.lane); (void)hipFree(kv.second.tile); (void)hipFree(kv.second.mask); }
    g_conv3_tables.clear();
  }
  dh_conv3::HostTables ht;   // pure integer host code (conv3_tables_host.h; swept under sanitizers by tests/test_conv_tables_host.py)
  if (const char* why = dh_conv3::build_tables<STRIDE, NT, WAVES, ESZ, MT, Conv3Params, HALF>(p, ncb, grid_override, &ht)) { dh::set_error("%s", why); return DH_EINVAL; }
  static_assert(sizeof(dh_conv3::TileDesc) == sizeof(int4), "tile descriptor = int4");
  const std::vector<int>& lane = ht.lane;
  const std::vector<dh_conv3::TileDesc>& tile = ht.tile;
  const std::vector<unsigned>& mask = ht.mask;
  Conv3Tables tb;
  DH_HIP(hipMalloc((void**)&tb.lane, lane.size() * sizeof(int)));
  DH_HIP(hipMalloc((void**)&tb.tile, tile.size() * sizeof(int4)));
  DH_HIP(hipMalloc((void**)&tb.mask, mask.size() * sizeof(unsigned)));
  DH_HIP(hipMemcpy(tb.lane, lane.data(), lane.size() * sizeof(int), hipMemcpyHostToDevice));
  DH_HIP(hipMemcpy(tb.tile, tile.data(), tile.size() * sizeof(int4), hipMemcpyHostToDevice));
  DH_HIP(hipMemcpy(tb.mask, mask.data(), mask.size() * sizeof(unsigned), hipMemcpyHostToDevice));
  g_conv3_tables[key] = tb;
  *out = tb;
  (void)groups_img;
  return DH_OK;
}

template <typename T, int STRIDE, int NT, int WAVES, bool DS = false, int MT = 2, bool WRES = false, int CLS = -1, bool HALF = false>
int launch_conv3x3_cfg(Conv3Params& p, const ConvLayer& L, hipStream_t st) {
  constexpr int STAGE_PX_BYTES = HALF ? CHUNK_BYTES / 2 : CHUNK_BYTES, CO_BLK = HALF ? 128 : 64;
  const int win_px = p.IMGS * p.HR * p.HP + p.WTAIL;
  const int win_bytes = win_px * STAGE_PX_BYTES;
  const size_t wslab = (size_t)(9 + (DS ? 1 : 0)) * SLAB_TAP, win_alloc = (win_bytes + 1023) & ~1023;
  const size_t nchunks = (size_t)L.cin * sizeof(T) / STAGE_PX_BYTES;
  // 2-deep ring of [weight slab | window] (WRES: all weight slabs once + ring of windows) + [2][scale|shift(|ds scale|ds shift)]
  const size_t lds = (WRES ? nchunks * wslab + 2 * win_alloc : 2 * (wslab + win_alloc)) + (HALF ? 4096 : DS ? 2048 : 1024);

  constexpr int MAXJ = (STRIDE == 2) ? (WAVES == 8 ? 5 : 10) : (NT == 2 ? 6 : 4);
  p.n_win_instr = HALF ? (win_px + 31) / 32 : (win_px + 15) / 16;
  DH_REQUIRE(p.n_win_instr <= MAXJ * WAVES, "conv3x3: staging window too large for the DMA plan");
  DH_REQUIRE(lds <= 160 * 1024, "conv3x3: LDS budget exceeded (%zu B)", lds);
  DH_REQUIRE(p.FIT ? p.IMGS * p.TH * p.TW <= (WAVES * MT / 2) * NT * 32 && STRIDE == 1 && !HALF
                   : p.IMGS * p.TH * p.TW == (HALF ? WAVES / 2 : WAVES * MT / 2) * NT * 32, "conv3x3: tile/pixel mismatch");
  DH_REQUIRE(L.cin * (int)sizeof(T) >= 2 * CHUNK_BYTES, "conv3x3: needs at least two channel chunks");
  const int groups = ((p.B + p.IMGS - 1) / p.IMGS) * p.tiles_y * p.tiles_x;
  p.ntiles = groups * (L.cout / CO_BLK);
  const int grid = std::min(256, p.ntiles);  // persistent: one workgroup per CU
  DH_REQUIRE(!WRES || grid % (L.cout / 64) == 0, "conv3x3: resident weights need a fixed cout block per workgroup");
  p.iters = (p.ntiles + grid - 1) / grid;
  Conv3Tables tb;
  int rc = conv3_tables<STRIDE, NT, WAVES, (int)sizeof(T), MT, HALF>(p, L.cout / CO_BLK, groups, &tb);
  if (rc) return rc;
  p.lane_tab = tb.lane; p.tile_tab = tb.tile; p.mask_tab = tb.mask;
  if constexpr (CLS >= 0) {   // parity class of a stride-2 data gradient: no cycle-stamped twin
    if ((rc = ensure_dyn_lds(reinterpret_cast<const void*>(&conv3x3_kernel<T, STRIDE, NT, WAVES, false, DS, MT, WRES, CLS>), 160 * 1024))) return rc;
    hipLaunchKernelGGL((conv3x3_kernel<T, STRIDE, NT, WAVES, false, DS, MT, WRES, CLS>), dim3(grid), dim3(WAVES * 64), lds, st, p);
    DH_LAUNCH_CHECK();
    return DH_OK;
  }
  if ((rc = ensure_dyn_lds(reinterpret_cast<const void*>(&conv3x3_kernel<T, STRIDE, NT, WAVES, false, DS, MT, WRES, -1, HALF>), 160 * 1024)) ||
      (rc = ensure_dyn_lds(reinterpret_cast<const void*>(&conv3x3_kernel<T, STRIDE, NT, WAVES, true, DS, MT, WRES, -1, HALF>), 160 * 1024))) return rc;
  if (p.stamps) hipLaunchKernelGGL((conv3x3_kernel<T, STRIDE, NT, WAVES, true, DS, MT, WRES, -1, HALF>), dim3(grid), dim3(WAVES * 64), lds, st, p);
  else hipLaunchKernelGGL((conv3x3_kernel<T, STRIDE, NT, WAVES, false, DS, MT, WRES, -1, HALF>), dim3(grid), dim3(WAVES * 64), lds, st, p);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

// Will a 3x3 / stride-2 conv + fused downsample of this shape run on the wide kernel (conv3x3.inc, HALF)?  Then its INPUT must be written in
// 16-channel planes by the conv before it (forward_impl decides with this before launching that conv).
inline bool s2_wide_eligible(const ConvLayer& c1, const ConvLayer& ds, int esz, int Wo) {
  static const bool on = dh::env_int("DH_CONV_S2_WIDE") != 0;
  return on && esz == 2 && c1.w2_dev && ds.w2_dev && c1.cout % 128 == 0 && Wo > 4;
}

// CLS >= 0: parity class (CLS >> 1, CLS & 1) of a stride-2 data gradient -- `in` = dZ [B][Hi][Wi][L.cin], Ho x Wo = the class's
// rows x columns, out = the FULL-SIZE dX [B][full_h][full_w][L.cout] (NHWC), of which the class owns pixels (2 i + py, 2 j + px).
template <typename T, int STRIDE, int CLS = -1>
int launch_conv3x3(const ConvLayer& L, const void* in, const void* res, void* out, int B, int Hi, int Wi,
                   bool relu, hipStream_t st, int Ho, int Wo, const ConvLayer* ds = nullptr, void* ds_out = nullptr,
                   bool blocked = false, int full_h = 0, int full_w = 0, bool in16 = false, bool out16 = false) {
  Conv3Params p;
  static_assert(CLS < 0 || STRIDE == 1, "parity classes run on the stride-1 kernel");
  DH_REQUIRE(blocked || (!in16 && !out16), "conv3x3: 16-channel planes are a variant of the channel-blocked layout");
  DH_REQUIRE(CLS < 0 || (!blocked && !ds && full_h > 0 && full_w > 0), "conv3x3: a parity class needs the NHWC layout and the size of dX");
  DH_REQUIRE(!blocked || sizeof(T) == 2, "conv3x3: the channel-blocked layout is the bf16 inference layout");
  if (blocked) {   // [image][C/32][H][W][32]
    p.in_px_bytes = CHUNK_BYTES; p.in_chunk_bytes = Hi * Wi * CHUNK_BYTES;
    p.out_px = 32; p.out_mt = Ho * Wo * 32; p.out_cb = 2 * p.out_mt;
  } else {         // NHWC
    p.in_px_bytes = L.cin * (int)sizeof(T); p.in_chunk_bytes = CHUNK_BYTES;
    p.out_px = L.cout; p.out_mt = 32; p.out_cb = 64;
  }
  if (CLS >= 0) {
    p.o_img = (int64_t)full_h * full_w * L.cout; p.o_row = 2 * full_w * L.cout; p.o_px = 2 * L.cout;
    p.o_base = ((CLS >> 1) * full_w + (CLS & 1)) * L.cout;
  } else {
    p.o_img = (int64_t)Ho * Wo * L.cout; p.o_row = Wo * p.out_px; p.o_px = p.out_px; p.o_base = 0;
  }
  // the residual shares the output's layout ...
  p.out_pr = 16; p.res_mt = p.out_mt; p.res_pr = 16; p.r_row = p.o_row; p.r_px = p.o_px; p.r_cb = p.out_cb; p.r_base = p.o_base;
  // ... unless this conv writes 16-channel planes [image][C/16][H][W][16] for the wide stride-2 kernel that follows (conv3x3.inc, HALF:
  // its half-chunk stages then read whole cache lines); the residual -- the block's input -- stays in 32-channel planes
  if (out16) {
    p.out_px = 16; p.out_pr = Ho * Wo * 16; p.out_mt = 2 * p.out_pr; p.out_cb = 2 * p.out_mt;
    p.o_row = Wo * 16; p.o_px = 16;
  }
  if (in16) { p.in_px_bytes = 32; p.in_chunk_bytes = Hi * Wi * 32; }
  p.ds_w = ds ? ds->w_dev : nullptr; p.ds_scale = ds ? ds->scale_dev : nullptr;
  p.ds_shift = ds ? ds->shift_dev : nullptr; p.ds_out = ds_out;
  p.in = in; p.w = L.w_dev; p.scale = L.scale_dev; p.shift = L.shift_dev; p.res = res; p.out = out;
  p.B = B; p.Hi = Hi; p.Wi = Wi; p.Cin = L.cin; p.Cout = L.cout; p.Ho = Ho; p.Wo = Wo;
  p.relu = relu ? 1 : 0;
  if (int zrc = zero_page(&p.zero_page)) return zrc;
  p.stamps = nullptr;
  if (g_stamps_on) {
    if (!g_stamps_dev) {
      DH_HIP(hipMalloc((void**)&g_stamps_dev, 64 * sizeof(unsigned long long)));
      DH_HIP(hipMemset(g_stamps_dev, 0, 64 * sizeof(unsigned long long)));
    }
    // one row of 8 counters per layer class: rows 0..3 stride 1 by cin 64..512, 4..6 stride 2
    const int row = (STRIDE == 1) ? (L.cin == 64 ? 0 : L.cin == 128 ? 1 : L.cin == 256 ? 2 : 3)
                                  : (L.cin == 64 ? 4 : L.cin == 128 ? 5 : 6);
    p.stamps = g_stamps_dev + 8 * row;
  }
  bool sample = false;
  // sampled: the dominant variant only (stride 1, 512-slot tiles: layers 1-3, and since round 5 layer 4 on its fit tiles)
  auto maybe_sample = [&](bool dominant) -> int {
    if (dominant && g_prof.on && g_prof.used + 2 <= g_prof.ev.size() && (g_prof.counter++ % g_prof.every) == 0) {
      sample = true;
      DH_HIP(hipEventRecord(g_prof.ev[g_prof.used], st));
    }
    return DH_OK;
  };
  int rc;
  if constexpr (STRIDE == 1) {
    p.HPH = 0;
    // 512-pixel tiles (NT = 2) when that still gives every CU a tile; small batches of small maps (training at
    // batch 64: 14x14 and 7x7) drop to 256- and 128-pixel tiles instead of leaving half the chip idle
    int variant;  // 0: NT=2 MT=2 (512 px)   1: NT=1 MT=2 (256 px)   2: NT=1 MT=1 (128 px)
    // candidates from the largest tile down; the first that gives every CU a tile wins, else the smallest (round 3: a launch
    // with few pixels -- a parity class of a stride-2 data gradient at batch 64 -- used to run 512-pixel tiles on half the chip)
    constexpr int min_tiles = 256;
    // DH_CONV_FIT=0: the round-4 candidate list (power-of-two tiles only) for A/B; default: fit tiles for 7 x 7 and 14 x 14 maps too
    static const bool fit_tiles = dh::env_int("DH_CONV_FIT") != 0;
    const dh_conv3::Cand cd = dh_conv3::pick_stride1(B, Ho, Wo, L.cout, min_tiles, fit_tiles && CLS < 0);   // conv3_tables_host.h
    dh_conv3::set_stride1_geometry(p, cd, Ho, Wo);
    variant = cd.variant;
    if ((rc = maybe_sample(variant == 0 && CLS < 0))) return rc;
    // weights resident in LDS when the layer has one cout block and its slabs fit beside the window ring (bf16 64 -> 64)
    const size_t wres_lds = (size_t)L.cin * sizeof(T) / CHUNK_BYTES * 9 * SLAB_TAP + 2 * (((size_t)(p.IMGS * p.HR * p.HP + p.WTAIL) * CHUNK_BYTES + 1023) & ~(size_t)1023) + 1024;
    if constexpr (CLS >= 0)
      rc = variant == 0 ? launch_conv3x3_cfg<T, 1, 2, 8, false, 2, false, CLS>(p, L, st)
         : variant == 1 ? launch_conv3x3_cfg<T, 1, 1, 8, false, 2, false, CLS>(p, L, st)
                        : launch_conv3x3_cfg<T, 1, 1, 8, false, 1, false, CLS>(p, L, st);
    else if (variant == 0 && sizeof(T) == 2 && wres_lds <= 160 * 1024)
      rc = launch_conv3x3_cfg<T, 1, 2, 8, false, 2, true>(p, L, st);
    else
    rc = variant == 0 ? launch_conv3x3_cfg<T, 1, 2, 8>(p, L, st)
       : variant == 1 ? launch_conv3x3_cfg<T, 1, 1, 8>(p, L, st)
                      : launch_conv3x3_cfg<T, 1, 1, 8, false, 1>(p, L, st);
  } else {
    // round 4: the wide variant (128 couts x 256 pixels per workgroup on half-chunk stages; conv3x3.inc, HALF) where its packing exists
    // (bf16 inference, fused downsample, cout % 128 == 0) and the map is at least 16 pixels wide.  DH_CONV_S2_WIDE=0: the 128-pixel kernel
    static const bool s2_wide = dh::env_int("DH_CONV_S2_WIDE") != 0;
    if constexpr (sizeof(T) == 2) {
      if (in16) {
        DH_REQUIRE(s2_wide && ds && blocked && L.w2_dev && ds->w2_dev && L.cout % 128 == 0 && dh_conv3::set_stride2_wide_geometry(p, Ho, Wo),
                   "conv3x3: an input in 16-channel planes needs the wide stride-2 kernel");
        p.w = L.w2_dev; p.ds_w = ds->w2_dev;
        p.out_cb = 4 * p.out_mt;   // a workgroup's cout block = four 32-cout planes
        return launch_conv3x3_cfg<T, 2, 2, 8, true, 2, false, -1, true>(p, L, st);
      }
    }
    dh_conv3::set_stride2_geometry(p, Ho, Wo);   // conv3_tables_host.h
    // 128-pixel tiles, 8 waves: wave pairs share pixels and split the 64 couts (two waves per SIMD)
    const size_t wres_lds = (size_t)L.cin * sizeof(T) / CHUNK_BYTES * (ds ? 10 : 9) * SLAB_TAP
                          + 2 * (((size_t)p.IMGS * p.HR * p.HP * CHUNK_BYTES + 1023) & ~(size_t)1023) + 2048;
    if (ds && sizeof(T) == 2 && wres_lds <= 160 * 1024) rc = launch_conv3x3_cfg<T, 2, 1, 8, true, 1, true>(p, L, st);
    else rc = ds ? launch_conv3x3_cfg<T, 2, 1, 8, true, 1>(p, L, st) : launch_conv3x3_cfg<T, 2, 1, 8, false, 1>(p, L, st);
  }
  if (rc) return rc;
  if (sample) {
    DH_HIP(hipEventRecord(g_prof.ev[g_prof.used + 1], st));
    g_prof.used += 2;
    g_prof.flops += 2.0 * B * Ho * Wo * (double)L.cout * 9 * L.cin;
  }
  return DH_OK;
}

// Data gradient of a 3x3 / stride-2 / pad-1 convolution WITHOUT the zero-upsampled copy of dZ: four stride-1 launches, one per
// parity class of the dX pixel (1, 2, 2 and 4 taps of the flipped + transposed operator `L.w_dev`; conv3x3.inc, CLS).  dz is
// [B][Ho][Wo][L.cin], dx (and res, the gradient joining from another branch, may be null) [B][Hi][Wi][L.cout], NHWC.
// (The classes write disjoint pixels and each fills at most half the chip at batch 64; running them on four streams, forked and
// joined by events, was measured SLOWER: float32 ResNet-18 step 9.64 -> 10.63 ms -- six cross-stream dependencies per convolution
// cost more than the idle CUs.  One stream.)
template <typename T, int NT, int WAVES, int MT>
int launch_dgrad_s2_merged_cfg(Conv3Params (&pc)[4], const int (&tiles_c)[4], const ConvLayer& L, hipStream_t st) {
  // pc[k]: the parameters of class order[k] (same tile shape, own Ho x Wo and output base); one launch over all of them
  static const int order[4] = {3, 1, 2, 0}, taps[4] = {4, 2, 2, 1};
  constexpr int MAXJ = NT == 2 ? 6 : 4;
  const Conv3Params& p0 = pc[0];
  const int win_bytes = p0.IMGS * p0.HR * p0.HP * CHUNK_BYTES;
  const size_t lds = 2 * ((size_t)9 * SLAB_TAP + ((win_bytes + 1023) & ~1023)) + 1024;
  const int n_win_instr = (p0.IMGS * p0.HR * p0.HP + 15) / 16;
  DH_REQUIRE(n_win_instr <= MAXJ * WAVES, "dgrad s2: staging window too large for the DMA plan");
  DH_REQUIRE(lds <= 160 * 1024, "dgrad s2: LDS budget exceeded (%zu B)", lds);
  DH_REQUIRE(p0.IMGS * p0.TH * p0.TW == (WAVES * MT / 2) * NT * 32, "dgrad s2: tile/pixel mismatch");
  DH_REQUIRE(L.cin * (int)sizeof(T) >= 2 * CHUNK_BYTES, "dgrad s2: needs at least two channel chunks");
  // workgroups per class: one at a time to the class whose (tiles per workgroup) x (taps) is largest
  int wg[4] = {0, 0, 0, 0}, total = 0;
  for (int k = 0; k < 4; ++k) if (tiles_c[k] > 0) { wg[k] = 1; ++total; }
  auto cost = [&](int k) { return wg[k] > 0 ? ((tiles_c[k] + wg[k] - 1) / wg[k]) * taps[k] : 0; };
  while (total < 256) {
    int best = -1;
    for (int k = 0; k < 4; ++k)
      if (wg[k] > 0 && wg[k] < tiles_c[k] && (best < 0 || cost(k) > cost(best))) best = k;
    if (best < 0) break;
    ++wg[best]; ++total;
  }
  S2ClassSched sch;
  sch.first[0] = 0;
  for (int k = 0; k < 4; ++k) {
    sch.first[k + 1] = sch.first[k] + wg[k];
    sch.iters[k] = 0; sch.lane[k] = nullptr; sch.tile[k] = nullptr; sch.mask[k] = nullptr;
    if (wg[k] == 0) continue;
    Conv3Params& p = pc[k];
    p.n_win_instr = n_win_instr;
    p.ntiles = tiles_c[k];
    p.iters = (tiles_c[k] + wg[k] - 1) / wg[k];
    Conv3Tables tb;
    int rc = conv3_tables<1, NT, WAVES, (int)sizeof(T), MT>(p, L.cout / 64, 0, &tb, wg[k]);
    if (rc) return rc;
    sch.iters[k] = p.iters; sch.lane[k] = tb.lane; sch.tile[k] = tb.tile; sch.mask[k] = tb.mask;
  }
  (void)order;
  Conv3Params pl = pc[0];
  pl.n_win_instr = n_win_instr;
  pl.lane_tab = nullptr; pl.tile_tab = nullptr; pl.mask_tab = nullptr; pl.stamps = nullptr;
  if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(&conv3x3_s2dgrad_kernel<T, NT, WAVES, MT>), 160 * 1024)) return rc;
  hipLaunchKernelGGL((conv3x3_s2dgrad_kernel<T, NT, WAVES, MT>), dim3(sch.first[4]), dim3(WAVES * 64), lds, st, pl, sch);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

// Data gradient of a 3x3 / stride-2 / pad-1 convolution WITHOUT the zero-upsampled copy of dZ: four stride-1 convolutions, one per
// parity class of the dX pixel (1, 2, 2 and 4 taps of the flipped + transposed operator `L.w_dev`; conv3x3.inc, CLS), side by side in ONE
// launch (conv3x3_s2dgrad_kernel).  dz is [B][Ho][Wo][L.cin], dx (and res, the gradient joining from another branch, may be null)
// [B][Hi][Wi][L.cout], NHWC.  (History: four launches on one stream -- each fills a fraction of the chip at batch 64; four streams forked
// and joined by events -- slower still, 9.64 -> 10.63 ms per float32 step.)
template <typename T>
int launch_dgrad_s2(const ConvLayer& L, const void* dz, const void* res, void* dx, int B, int Ho, int Wo, int Hi, int Wi, hipStream_t st) {
  DH_REQUIRE(Ho == (Hi + 2 - 3) / 2 + 1 && Wo == (Wi + 2 - 3) / 2 + 1, "dgrad s2: %dx%d is not the stride-2 output of %dx%d", Ho, Wo, Hi, Wi);
  int rc;
  const int re = (Hi + 1) / 2, ro = Hi / 2, ce = (Wi + 1) / 2, co = Wi / 2;   // rows / columns of dX with even / odd index
  if (Hi % 2 == 0 && Wi % 2 == 0) {
    // all four classes from one staged dZ window (conv3x3.inc, CLS == 4): a stride-1 convolution over the dZ grid with four accumulator
    // sets; 256- or 128-pixel tiles (NT = 1)
    Conv3Params p;
    p.in_px_bytes = L.cin * (int)sizeof(T); p.in_chunk_bytes = CHUNK_BYTES;
    p.out_px = L.cout; p.out_mt = 32; p.out_cb = 64;
    p.o_img = (int64_t)Hi * Wi * L.cout; p.o_row = 2 * Wi * L.cout; p.o_px = 2 * L.cout; p.o_base = 0;
    p.out_pr = 16; p.res_mt = p.out_mt; p.res_pr = 16; p.r_row = p.o_row; p.r_px = p.o_px; p.r_cb = p.out_cb; p.r_base = 0;
    for (int c = 0; c < 4; ++c) p.cls_base[c] = ((c >> 1) * Wi + (c & 1)) * L.cout;
    p.ds_w = nullptr; p.ds_scale = nullptr; p.ds_shift = nullptr; p.ds_out = nullptr;
    p.in = dz; p.w = L.w_dev; p.scale = L.scale_dev; p.shift = L.shift_dev; p.res = res; p.out = dx;
    p.B = B; p.Hi = Ho; p.Wi = Wo; p.Cin = L.cin; p.Cout = L.cout; p.Ho = re; p.Wo = ce;
    p.relu = 0; p.stamps = nullptr; p.HPH = 0;
    if ((rc = zero_page(&p.zero_page))) return rc;
    struct CandA { int th, tw, imgs, hp, mt; };
    CandA ca[2];
    if (ce > 8) { ca[0] = {16, 16, 1, 18, 2}; ca[1] = {8, 8, 2, 12, 1}; }
    else { ca[0] = {8, 8, 4, 12, 2}; ca[1] = {8, 8, 2, 12, 1}; }
    auto ntiles = [&](const CandA& c) { return ((B + c.imgs - 1) / c.imgs) * ((re + c.th - 1) / c.th) * ((ce + c.tw - 1) / c.tw) * (L.cout / 64); };
    const CandA& cd = ntiles(ca[0]) >= 256 ? ca[0] : ca[1];
    p.TH = cd.th; p.TW = cd.tw; p.IMGS = cd.imgs; p.HP = cd.hp; p.HR = cd.th + 2; p.HC = cd.tw + 2;
    p.tiles_y = (re + cd.th - 1) / cd.th; p.tiles_x = (ce + cd.tw - 1) / cd.tw;
    return cd.mt == 2 ? launch_conv3x3_cfg<T, 1, 1, 8, false, 2, false, 4>(p, L, st) : launch_conv3x3_cfg<T, 1, 1, 8, false, 1, false, 4>(p, L, st);
  }
  // one tile shape for all classes, picked on the largest (even, even) class: the largest tile that still gives the launch 256 tiles
  static const int order[4] = {3, 1, 2, 0};
  const int rows_of[4] = {re, re, ro, ro}, cols_of[4] = {ce, co, ce, co};   // by class id (py, px) = (id >> 1, id & 1)
  struct Cand { int th, tw, imgs, hp, variant; };
  Cand cands[4];
  int nc = 0;
  if (ce > 16) {
    const int slots_a = ((re + 15) / 16) * ((ce + 31) / 32), slots_b = ((re + 7) / 8) * ((ce + 63) / 64);
    if (slots_b < slots_a) cands[nc++] = {8, 64, 1, 66, 0};
    else cands[nc++] = {16, 32, 1, 34, 0};
    cands[nc++] = {16, 16, 1, 18, 1};
  } else if (ce > 8) {
    cands[nc++] = {16, 16, 2, 18, 0};
    cands[nc++] = {16, 16, 1, 18, 1};
    cands[nc++] = {8, 8, 2, 12, 2};
  } else {
    cands[nc++] = {8, 8, 4, 12, 1};
    cands[nc++] = {8, 8, 2, 12, 2};
  }
  auto tiles_of = [&](const Cand& c, int rows, int cols) {
    return rows > 0 && cols > 0 ? ((B + c.imgs - 1) / c.imgs) * ((rows + c.th - 1) / c.th) * ((cols + c.tw - 1) / c.tw) * (L.cout / 64) : 0;
  };
  int pick = nc - 1;
  for (int i = 0; i < nc; ++i) {
    int tot = 0;
    for (int id = 0; id < 4; ++id) tot += tiles_of(cands[i], rows_of[id], cols_of[id]);
    if (tot >= 256) { pick = i; break; }
  }
  const Cand& cd = cands[pick];
  Conv3Params pc[4];
  int tiles_c[4];
  const void* zp = nullptr;
  if (int zrc = zero_page(&zp)) return zrc;
  for (int k = 0; k < 4; ++k) {
    const int id = order[k], rows = rows_of[id], cols = cols_of[id];
    Conv3Params& p = pc[k];
    p.in_px_bytes = L.cin * (int)sizeof(T); p.in_chunk_bytes = CHUNK_BYTES;
    p.out_px = L.cout; p.out_mt = 32; p.out_cb = 64;
    p.o_img = (int64_t)Hi * Wi * L.cout; p.o_row = 2 * Wi * L.cout; p.o_px = 2 * L.cout;
    p.o_base = ((id >> 1) * Wi + (id & 1)) * L.cout;
    p.out_pr = 16; p.res_mt = p.out_mt; p.res_pr = 16; p.r_row = p.o_row; p.r_px = p.o_px; p.r_cb = p.out_cb; p.r_base = p.o_base;
    p.ds_w = nullptr; p.ds_scale = nullptr; p.ds_shift = nullptr; p.ds_out = nullptr;
    p.in = dz; p.w = L.w_dev; p.scale = L.scale_dev; p.shift = L.shift_dev; p.res = res; p.out = dx;
    p.B = B; p.Hi = Ho; p.Wi = Wo; p.Cin = L.cin; p.Cout = L.cout; p.Ho = rows; p.Wo = cols;
    p.relu = 0; p.zero_page = zp; p.stamps = nullptr;
    p.HPH = 0; p.TH = cd.th; p.TW = cd.tw; p.IMGS = cd.imgs; p.HP = cd.hp; p.HR = cd.th + 2; p.HC = cd.tw + 2;
    p.tiles_y = (rows + cd.th - 1) / cd.th; p.tiles_x = (cols + cd.tw - 1) / cd.tw;
    p.lane_tab = nullptr; p.tile_tab = nullptr; p.mask_tab = nullptr; p.ntiles = 0; p.iters = 0; p.n_win_instr = 0;
    tiles_c[k] = tiles_of(cd, rows, cols);
  }
  return cd.variant == 0 ? launch_dgrad_s2_merged_cfg<T, 2, 8, 2>(pc, tiles_c, L, st)
       : cd.variant == 1 ? launch_dgrad_s2_merged_cfg<T, 1, 8, 2>(pc, tiles_c, L, st)
                         : launch_dgrad_s2_merged_cfg<T, 1, 8, 1>(pc, tiles_c, L, st);
}

template <typename T>
int run_conv(const ConvLayer& L, const void* in, const void* res, void* out, int B, int Hi, int Wi,
             bool relu, hipStream_t st, int* Ho_out, int* Wo_out, bool blocked = false, bool out16 = false) {
  const int pad = L.ks / 2;
  const int Ho = (Hi + 2 * pad - L.ks) / L.stride + 1, Wo = (Wi + 2 * pad - L.ks) / L.stride + 1;
  *Ho_out = Ho; *Wo_out = Wo;
  DH_REQUIRE((int64_t)B * Hi * Wi * L.cin * (int64_t)sizeof(T) < ((int64_t)1 << 32),
             "conv %s: input larger than 4 GiB, reduce the batch", L.name.c_str());
  if (L.ks == 3 && L.stride == 1) return launch_conv3x3<T, 1>(L, in, res, out, B, Hi, Wi, relu, st, Ho, Wo, nullptr, nullptr, blocked, 0, 0, false, out16);
  DH_REQUIRE(!out16, "conv %s: only the stride-1 3x3 kernel writes 16-channel planes", L.name.c_str());
  if (L.ks == 3 && L.stride == 2) return launch_conv3x3<T, 2>(L, in, res, out, B, Hi, Wi, relu, st, Ho, Wo, nullptr, nullptr, blocked);
  DH_REQUIRE(!blocked, "conv %s: only the 3x3 kernels read the channel-blocked layout", L.name.c_str());
  ConvParams p;
  p.in = in; p.w = L.w_dev; p.scale = L.scale_dev; p.shift = L.shift_dev; p.res = res; p.out = out;
  p.B = B; p.Hi = Hi; p.Wi = Wi; p.Cin = L.cin; p.Cout = L.cout; p.Ho = Ho; p.Wo = Wo;
  if (p.Ho > 8 || p.Wo > 8) { p.TH = 16; p.TW = 16; p.IMGS = 1; }
  else { p.TH = 8; p.TW = 8; p.IMGS = 4; }
  p.tiles_y = (p.Ho + p.TH - 1) / p.TH; p.tiles_x = (p.Wo + p.TW - 1) / p.TW;
  p.relu = relu ? 1 : 0;
  if (L.ks == 1 && L.stride == 2) return launch_conv<T, 1, 2, false>(p, st);
  dh::set_error("conv %s: unsupported shape", L.name.c_str());
  return DH_EINVAL;
}

// fused bf16 stem (conv1 + bn1 + relu + maxpool) of `B` tiles -> channel-blocked [image][2][H2][W2][32] bf16
int launch_stem_pool(dh_resnet18* net, const float* x, const uint8_t* slide, int64_t slide_h, int64_t slide_w,
                     const int32_t* yx, int B, int P, void* out, hipStream_t st) {
  const int H1 = (P + 6 - 7) / 2 + 1, H2 = (H1 + 2 - 3) / 2 + 1;
  StemPoolParams sp;
  DH_REQUIRE(!slide || slide_w < (1 << 24), "stem: slide rows of %lld pixels are not supported (< 2^24)", (long long)slide_w);
  sp.x_nchw = x; sp.slide = slide; sp.yx = yx; sp.row_bytes = slide_w * 3; sp.slide_bytes = slide_h * slide_w * 3;
  sp.w = net->convs[0].w_dev; sp.scale = net->convs[0].scale_dev; sp.shift = net->convs[0].shift_dev; sp.out = out;
  sp.B = B; sp.P = P; sp.Hc = H1; sp.Wc = H1; sp.Hp = H2; sp.Wp = H2;
  sp.tiles_y = (H2 + SP_PR - 1) / SP_PR; sp.tiles_x = (H2 + SP_PC - 1) / SP_PC;
  sp.nstrips = B * sp.tiles_x;                 // a strip = one image x 15 pooled columns, swept top to bottom
  const int grid = std::min(768, sp.nstrips);  // persistent: three 4-wave workgroups per CU
  sp.iters = ((sp.nstrips + grid - 1) / grid) * sp.tiles_y;
  int arc;
  if ((arc = ensure_dyn_lds(reinterpret_cast<const void*>(&stem_pool_kernel<true, false>), SP_LDS)) ||
      (arc = ensure_dyn_lds(reinterpret_cast<const void*>(&stem_pool_kernel<true, true>), SP_LDS)) ||
      (arc = ensure_dyn_lds(reinterpret_cast<const void*>(&stem_pool_kernel<false, false>), SP_LDS))) return arc;
  sp.stamps = nullptr;
  if (g_stamps_on && slide) {
    if (!g_stamps_dev) {
      DH_HIP(hipMalloc((void**)&g_stamps_dev, 64 * sizeof(unsigned long long)));
      DH_HIP(hipMemset(g_stamps_dev, 0, 64 * sizeof(unsigned long long)));
    }
    sp.stamps = g_stamps_dev + 8 * 7;
  }
  if (slide && sp.stamps) hipLaunchKernelGGL((stem_pool_kernel<true, true>), dim3(grid), dim3(SP_T), SP_LDS, st, sp);
  else if (slide) hipLaunchKernelGGL((stem_pool_kernel<true, false>), dim3(grid), dim3(SP_T), SP_LDS, st, sp);
  else hipLaunchKernelGGL((stem_pool_kernel<false, false>), dim3(grid), dim3(SP_T), SP_LDS, st, sp);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

template <typename T>
int forward_impl(dh_resnet18* net, const float* x, const uint8_t* slide, int64_t slide_h, int64_t slide_w,
                 const int32_t* yx, int64_t n64, int P, float* logits, hipStream_t st) {
  const int B = (int)n64;
  const int esz = (int)sizeof(T);
  const int H1 = (P + 6 - 7) / 2 + 1;       // stem out
  const int H2 = (H1 + 2 - 3) / 2 + 1;      // pool out
  const size_t stem_bytes = (size_t)B * H1 * H1 * 64 * esz;
  const size_t act_bytes = (size_t)B * H2 * H2 * 64 * esz;  // largest post-pool activation
  // BEFORE any launch: the conv schedule tables carry window byte offsets as 32-bit words (the stem and pool kernels index with
  // 64 bits) -- a launch whose largest conv input does not fit is refused here, not half-way through the network
  DH_REQUIRE(act_bytes < ((size_t)1 << 32),
             "resnet18 forward: %d tiles of %d x %d in %s make an activation larger than 4 GiB: use fewer tiles per launch", B, P, P,
             sizeof(T) == 2 ? "bf16" : "float32");
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t need = al(stem_bytes) + 3 * al(act_bytes);
  if (need > net->ws_bytes) {
    if (net->ws) DH_HIP(hipFree(net->ws));
    net->ws = nullptr; net->ws_bytes = 0;
    DH_HIP(hipMalloc(&net->ws, need));
    net->ws_bytes = need;
  }
  char* base = static_cast<char*>(net->ws);
  void* S = base;
  void* bufA = base + al(stem_bytes);
  void* bufB = base + al(stem_bytes) + al(act_bytes);
  void* bufC = base + al(stem_bytes) + 2 * al(act_bytes);

  if constexpr (sizeof(T) == 2) {
    // bf16: fused stem + BN + ReLU + maxpool (persistent, weights resident in LDS), straight into bufA
    if (int rc = launch_stem_pool(net, x, slide, slide_h, slide_w, yx, B, P, bufA, st)) return rc;
  } else {
  // stem
  {
    StemParams sp;
    sp.x_nchw = x; sp.slide = slide; sp.yx = yx; sp.row_bytes = slide_w * 3;
    sp.w = net->convs[0].w_dev; sp.scale = net->convs[0].scale_dev; sp.shift = net->convs[0].shift_dev;
    sp.out = S; sp.B = B; sp.P = P; sp.Ho = H1; sp.Wo = H1; sp.relu = 1;
    sp.tiles_y = (H1 + STEM_TH - 1) / STEM_TH; sp.tiles_x = (H1 + STEM_TW - 1) / STEM_TW;
    const size_t wb = esz == 2 ? (size_t)7 * 2 * 2 * FRAG_BYTES : (size_t)7 * 11 * 2 * 256;
    const size_t lds = wb + (size_t)STEM_ROWS * STEM_ROWE * esz;
    const int grid = B * sp.tiles_y * sp.tiles_x;
    if (slide) {
      if (int arc = ensure_dyn_lds(reinterpret_cast<const void*>(&stem_kernel<T, true>), 96 * 1024)) return arc;
      hipLaunchKernelGGL((stem_kernel<T, true>), dim3(grid), dim3(256), lds, st, sp);
    } else {
      if (int arc = ensure_dyn_lds(reinterpret_cast<const void*>(&stem_kernel<T, false>), 96 * 1024)) return arc;
      hipLaunchKernelGGL((stem_kernel<T, false>), dim3(grid), dim3(256), lds, st, sp);
    }
    DH_LAUNCH_CHECK();
  }
  // maxpool
  {
    const int64_t total = (int64_t)B * H2 * H2 * (64 / (16 / esz));
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL((maxpool_kernel<T>), dim3(grid), dim3(256), 0, st, static_cast<const T*>(S),
                       static_cast<T*>(bufA), B, H1, H1, 64, H2, H2);
    DH_LAUNCH_CHECK();
  }
  }
  // residual stages: X lives in bufA; T in bufB; downsample in bufC.  bf16 activations are channel-blocked
  // ([image][C/32][H][W][32], written that way by the fused stem), f32 activations NHWC.
  constexpr bool BLK = sizeof(T) == 2;
  int H = H2, W = H2;
  size_t ci = 1;
  bool x16 = false;   // X holds the block input in 16-channel planes (written that way for the wide stride-2 kernel)
  void* X = bufA;     // block input / output; bufB: the block's intermediate; `other`: the third buffer
  for (int s = 0; s < 4; ++s) {
    for (int blk = 0; blk < 2; ++blk) {
      const bool has_ds = (blk == 0 && s > 0);
      const ConvLayer& c1 = net->convs[ci];
      const ConvLayer& c2 = net->convs[ci + 1];
      void* other = X == bufA ? bufC : bufA;
      int Ho, Wo, h2, w2;
      int rc;
      const void* resid = X;
      if (has_ds) {
        // conv1 (3x3/2 + BN + ReLU) and the 1x1/2 downsample (+ BN) of the block in ONE launch
        Ho = (H + 2 - 3) / 2 + 1; Wo = (W + 2 - 3) / 2 + 1;
        rc = launch_conv3x3<T, 2>(c1, X, nullptr, bufB, B, H, W, true, st, Ho, Wo, &net->convs[ci + 2], other, BLK, 0, 0, x16, false);
        if (rc) return rc;
        resid = other;
      } else {
        DH_REQUIRE(!x16, "resnet18 forward: only a downsample block reads 16-channel planes");
        rc = run_conv<T>(c1, X, nullptr, bufB, B, H, W, true, st, &Ho, &Wo, BLK);
        if (rc) return rc;
      }
      // the block that follows starts with a stride-2 conv on the wide kernel: this block's output goes out in 16-channel planes
      x16 = false;
      if (blk == 1 && s < 3) {
        const ConvLayer& n1 = net->convs[ci + 2];
        const ConvLayer& nds = net->convs[ci + 4];
        x16 = BLK && s2_wide_eligible(n1, nds, esz, (Wo + 2 - 3) / 2 + 1);
      }
      // conv2 + residual.  Same layout: in place over the residual (every lane reads the elements it then writes).  16-channel planes out,
      // 32-channel planes in: the two layouts put different pixels at the same address -- the output goes to the third buffer
      void* O = X;
      if (has_ds) O = X;                 // the block input is dead once conv1 and the downsample have run (stream order)
      else if (x16) O = other;
      rc = run_conv<T>(c2, bufB, resid, O, B, Ho, Wo, true, st, &h2, &w2, BLK, x16);
      if (rc) return rc;
      X = O;
      H = Ho; W = Wo;
      ci += has_ds ? 3 : 2;
    }
  }
  hipLaunchKernelGGL((avgpool_fc_kernel<T>), dim3(B), dim3(256), 0, st, static_cast<const T*>(X), H * W,
                     512, BLK ? 1 : 0, net->fc_w_dev, net->fc_b_dev, net->n_classes, logits);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

template <typename V>
int upload(const std::vector<V>& v, void** dev) {
  if (*dev) { DH_HIP(hipFree(*dev)); *dev = nullptr; }
  DH_HIP(hipMalloc(dev, v.size() * sizeof(V)));
  DH_HIP(hipMemcpy(*dev, v.data(), v.size() * sizeof(V), hipMemcpyHostToDevice));
  return DH_OK;
}

}  // namespace

extern "C" int dh_resnet18_create(dh_resnet18** out, int32_t n_classes, int32_t dtype) {
  DH_REQUIRE(out != nullptr, "resnet18 create: null output");
  DH_REQUIRE(n_classes > 0 && n_classes <= 1024, "resnet18 create: n_classes=%d", n_classes);
  DH_REQUIRE(dtype == DH_DTYPE_F32 || dtype == DH_DTYPE_BF16, "resnet18 create: bad dtype %d", dtype);
  if (int erc = dh::env_check()) return erc;   // a mistyped DH_* switch stops here, named by dh_last_error()
  auto* net = new dh_resnet18();
  net->n_classes = n_classes;
  net->dtype = dtype;
  build_topology(net);
  *out = net;
  return DH_OK;
}

extern "C" int dh_resnet18_train_end(dh_resnet18* net);
extern "C" void dh_resnet18_destroy(dh_resnet18* net) {
  if (!net) return;
  if (net->train) (void)dh_resnet18_train_end(net);
  for (auto& c : net->convs) {
    if (c.w_dev) (void)hipFree(c.w_dev);
    if (c.w2_dev) (void)hipFree(c.w2_dev);
    if (c.scale_dev) (void)hipFree(c.scale_dev);
    if (c.shift_dev) (void)hipFree(c.shift_dev);
  }
  if (net->fc_w_dev) (void)hipFree(net->fc_w_dev);
  if (net->fc_b_dev) (void)hipFree(net->fc_b_dev);
  if (net->ws) (void)hipFree(net->ws);
  delete net;
}

extern "C" int dh_resnet18_set_param(dh_resnet18* net, const char* name, const float* data,
                                     int64_t n_elem) {
  DH_REQUIRE(net && name && data, "resnet18 set_param: null argument");
  const int64_t want = expected_elems(net, name);
  DH_REQUIRE(want >= 0, "resnet18 set_param: unknown parameter '%s'", name);
  DH_REQUIRE(want == n_elem, "resnet18 set_param: '%s' has %lld elements, expected %lld", name,
             (long long)n_elem, (long long)want);
  net->params[name].assign(data, data + n_elem);
  net->finalized = false;
  return DH_OK;
}

extern "C" int dh_resnet18_finalize(dh_resnet18* net, void* stream) {
  DH_REQUIRE(net != nullptr, "resnet18 finalize: null handle");
  (void)stream;
  auto get = [&](const std::string& k) -> const std::vector<float>* {
    auto it = net->params.find(k);
    return it == net->params.end() ? nullptr : &it->second;
  };
  const int esz = net->esz();
  for (size_t i = 0; i < net->convs.size(); ++i) {
    ConvLayer& c = net->convs[i];
    const auto* w = get(c.name + ".weight");
    const auto *g = get(c.bn + ".weight"), *b = get(c.bn + ".bias"), *m = get(c.bn + ".running_mean"),
               *v = get(c.bn + ".running_var");
    DH_REQUIRE(w && g && b && m && v, "resnet18 finalize: parameters of '%s' / '%s' are not all set",
               c.name.c_str(), c.bn.c_str());
    std::vector<uint8_t> packed;
    if (i == 0 && esz == 2) pack_stem_pool_weights(w->data(), packed);   // bf16 inference runs the fused stem only
    else if (i == 0) pack_stem_weights(w->data(), esz, packed);
    else pack_conv_weights(w->data(), c.cout, c.cin, c.ks, esz, packed);
    int rc = upload(packed, &c.w_dev);
    if (rc) return rc;
    // the wide stride-2 variant's packing: a 3x3 / stride-2 conv of a downsample block and that block's 1x1 / stride-2 conv (bf16 inference)
    const bool wide3 = esz == 2 && c.ks == 3 && c.stride == 2 && c.cout % 128 == 0 && c.cin % 16 == 0;
    const bool wide1 = esz == 2 && c.ks == 1 && c.stride == 2 && c.cout % 128 == 0 && c.cin % 16 == 0;
    if (wide3 || wide1) {
      std::vector<uint8_t> packed2;
      pack_conv_weights_wide(w->data(), c.cout, c.cin, c.ks, packed2);
      if ((rc = upload(packed2, &c.w2_dev))) return rc;
    }
    // eval-mode BN (eps = 1e-5, torch default): y = x*scale + shift
    std::vector<float> sc(c.cout), sh(c.cout);
    for (int k = 0; k < c.cout; ++k) {
      const double s = (double)(*g)[k] / sqrt((double)(*v)[k] + 1e-5);
      sc[k] = (float)s;
      sh[k] = (float)((double)(*b)[k] - (double)(*m)[k] * s);
    }
    rc = upload(sc, reinterpret_cast<void**>(&c.scale_dev));
    if (rc) return rc;
    rc = upload(sh, reinterpret_cast<void**>(&c.shift_dev));
    if (rc) return rc;
  }
  const auto *fw = get("fc.weight"), *fb = get("fc.bias");
  DH_REQUIRE(fw && fb, "resnet18 finalize: fc.weight / fc.bias are not set");
  int rc = upload(*fw, reinterpret_cast<void**>(&net->fc_w_dev));
  if (rc) return rc;
  rc = upload(*fb, reinterpret_cast<void**>(&net->fc_b_dev));
  if (rc) return rc;
  net->finalized = true;
  return DH_OK;
}

static int check_forward_args(dh_resnet18* net, int64_t n, int32_t P, const void* logits) {
  DH_REQUIRE(net && (logits || n == 0), "resnet18 forward: null argument");
  DH_REQUIRE(net->finalized, "resnet18 forward: call dh_resnet18_finalize after setting parameters");
  DH_REQUIRE(n >= 0 && n <= 4096, "resnet18 forward: batch %lld out of range [0, 4096]", (long long)n);
  DH_REQUIRE(P >= 32 && P <= 1024, "resnet18 forward: patch %d out of range [32, 1024]", P);
  return DH_OK;
}

extern "C" int dh_resnet18_forward(dh_resnet18* net, const float* x, int64_t n, int32_t P,
                                   float* logits, void* stream) {
  int rc = check_forward_args(net, n, P, logits);
  if (rc) return rc;
  if (n == 0) return DH_OK;
  DH_REQUIRE(x != nullptr, "resnet18 forward: null input");
  hipStream_t st = dh::as_stream(stream);
  return net->dtype == DH_DTYPE_F32
             ? forward_impl<float>(net, x, nullptr, 0, 0, nullptr, n, P, logits, st)
             : forward_impl<__bf16>(net, x, nullptr, 0, 0, nullptr, n, P, logits, st);
}

extern "C" int dh_resnet18_forward_tiles(dh_resnet18* net, const uint8_t* slide, int64_t h, int64_t w,
                                         const int32_t* yx, int64_t n, int32_t P, float* logits,
                                         void* stream) {
  int rc = check_forward_args(net, n, P, logits);
  if (rc) return rc;
  if (n == 0) return DH_OK;
  DH_REQUIRE(slide && yx, "resnet18 forward_tiles: null slide or origins");
  DH_REQUIRE(h >= P && w >= P, "resnet18 forward_tiles: patch %d does not fit %lldx%lld", P,
             (long long)h, (long long)w);
  hipStream_t st = dh::as_stream(stream);
  return net->dtype == DH_DTYPE_F32
             ? forward_impl<float>(net, nullptr, slide, h, w, yx, n, P, logits, st)
             : forward_impl<__bf16>(net, nullptr, slide, h, w, yx, n, P, logits, st);
}

// ---------------------------------------------------------------------------
// Test hooks (declared in include/deephisto_hip.h under "debug"): run one conv
// layer on caller-provided NHWC data, and read back the stem activation of the
// last forward.  They exist so the GPU parity tests can localise an error to a
// kernel; the product path never calls them.
// ---------------------------------------------------------------------------
namespace {
template <typename T>
__global__ void to_f32_kernel(const T* __restrict__ in, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if constexpr (sizeof(T) == 2) out[i] = bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(in)[i]);
    else out[i] = in[i];
  }
}
}  // namespace

extern "C" int dh_debug_conv_bn_act(const void* in_dev, const float* w_host, const float* scale_host,
                                    const float* shift_host, const void* res_dev, void* out_dev,
                                    int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout,
                                    int32_t ks, int32_t stride, int32_t relu, int32_t dtype,
                                    void* stream) {
  DH_REQUIRE(in_dev && w_host && scale_host && shift_host && out_dev, "debug conv: null pointer");
  DH_REQUIRE(dtype == DH_DTYPE_F32 || dtype == DH_DTYPE_BF16, "debug conv: bad dtype");
  const int esz = dtype == DH_DTYPE_F32 ? 4 : 2;
  DH_REQUIRE(cout % 64 == 0 && cin % (CHUNK_BYTES / esz) == 0, "debug conv: channel counts");
  ConvLayer L{"debug", "debug", cin, cout, ks, stride};
  std::vector<uint8_t> packed;
  pack_conv_weights(w_host, cout, cin, ks, esz, packed);
  int rc = upload(packed, &L.w_dev);
  if (rc) return rc;
  std::vector<float> sc(scale_host, scale_host + cout), sh(shift_host, shift_host + cout);
  rc = upload(sc, reinterpret_cast<void**>(&L.scale_dev));
  if (!rc) rc = upload(sh, reinterpret_cast<void**>(&L.shift_dev));
  int ho, wo;
  hipStream_t st = dh::as_stream(stream);
  if (!rc)
    rc = dtype == DH_DTYPE_F32 ? run_conv<float>(L, in_dev, res_dev, out_dev, B, H, W, relu != 0, st, &ho, &wo)
                               : run_conv<__bf16>(L, in_dev, res_dev, out_dev, B, H, W, relu != 0, st, &ho, &wo);
  hipError_t e = hipStreamSynchronize(st);
  if (L.w_dev) (void)hipFree(L.w_dev);
  if (L.scale_dev) (void)hipFree(L.scale_dev);
  if (L.shift_dev) (void)hipFree(L.shift_dev);
  if (!rc && e != hipSuccess) { dh::set_error("debug conv: %s", hipGetErrorString(e)); rc = DH_EHIP; }
  return rc;
}

extern "C" int dh_debug_stem_out(dh_resnet18* net, int64_t n, int32_t P, float* out_dev, void* stream) {
  DH_REQUIRE(net && out_dev && net->ws, "debug stem out: no forward has run");
  const int H1 = (P + 6 - 7) / 2 + 1;
  const int64_t elems = n * H1 * H1 * 64;
  DH_REQUIRE((size_t)elems * net->esz() <= net->ws_bytes, "debug stem out: workspace smaller than request");
  hipStream_t st = dh::as_stream(stream);
  if (net->dtype == DH_DTYPE_F32)
    hipLaunchKernelGGL((to_f32_kernel<float>), dim3(1024), dim3(256), 0, st, static_cast<const float*>(net->ws), out_dev, elems);
  else
    hipLaunchKernelGGL((to_f32_kernel<__bf16>), dim3(1024), dim3(256), 0, st, static_cast<const __bf16*>(net->ws), out_dev, elems);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

namespace {
__global__ void blocked_to_nhwc_f32_kernel(const __bf16* in, float* out, int64_t n, int hw) {   // [n][2][hw][32] -> [n][hw][64]
  const int64_t total = n * hw * 64;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % 64);
    const int64_t px = (i / 64) % hw, b = i / 64 / hw;
    out[i] = (float)in[((b * 2 + c / 32) * hw + px) * 32 + c % 32];
  }
}
}  // namespace

extern "C" int dh_debug_stem_pool_bf16(dh_resnet18* net, const uint8_t* slide_dev, int64_t slide_h, int64_t slide_w,
                                       const int32_t* yx_dev, int64_t n, int32_t P, float* out_dev, void* stream) {
  DH_REQUIRE(net && net->finalized && net->dtype == DH_DTYPE_BF16, "debug stem pool: needs a finalized bf16 network");
  DH_REQUIRE(slide_dev && yx_dev && out_dev && n > 0 && n <= (1 << 20) && P >= 32 && P <= 1024, "debug stem pool: bad arguments");
  hipStream_t st = dh::as_stream(stream);
  const int H1 = (P + 6 - 7) / 2 + 1, H2 = (H1 + 2 - 3) / 2 + 1;
  void* tmp = nullptr;
  DH_HIP(hipMalloc(&tmp, (size_t)n * H2 * H2 * 64 * 2));
  int rc = launch_stem_pool(net, nullptr, slide_dev, slide_h, slide_w, yx_dev, (int)n, P, tmp, st);
  if (!rc) {
    hipLaunchKernelGGL(blocked_to_nhwc_f32_kernel, dim3(1024), dim3(256), 0, st, static_cast<const __bf16*>(tmp), out_dev, n, H2 * H2);
    if (hipGetLastError() != hipSuccess) rc = DH_EHIP;
    (void)hipStreamSynchronize(st);
  }
  (void)hipFree(tmp);
  return rc;
}

// ---------------------------------------------------------------------------
// dominant-kernel timing (see Profiler above)
// ---------------------------------------------------------------------------
extern "C" int dh_profile_start(int32_t sample_every, int32_t max_samples) {
  DH_REQUIRE(sample_every > 0 && max_samples > 0 && max_samples <= (1 << 20), "profile start: bad arguments");
  for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.assign((size_t)max_samples * 2, nullptr);
  for (auto& e : g_prof.ev) DH_HIP(hipEventCreate(&e));
  g_prof.every = sample_every; g_prof.counter = 0; g_prof.used = 0; g_prof.flops = 0.0;
  g_prof.on = true;
  return DH_OK;
}

extern "C" int dh_profile_stop(double* total_ms, double* total_flops, int64_t* n_samples) {
  g_prof.on = false;
  double ms = 0.0;
  for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
    DH_HIP(hipEventSynchronize(g_prof.ev[i + 1]));
    float t = 0.f;
    DH_HIP(hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]));
    ms += t;
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = g_prof.flops;
  if (n_samples) *n_samples = (int64_t)(g_prof.used / 2);
  for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.clear(); g_prof.used = 0;
  return DH_OK;
}

// Diagnostic phase stamps of the 3x3 conv kernel (DH tests/tools only): enable, then read
// 8 rows x 8 counters {load-issue, mfma, barrier, epilogue, lds-write, barrier, workgroups, -}.
extern "C" int dh_debug_stamps(int32_t enable, unsigned long long* out64_host) {
  g_stamps_on = enable != 0;
  if (out64_host) {
    if (!g_stamps_dev) { memset(out64_host, 0, 64 * sizeof(unsigned long long)); return DH_OK; }
    DH_HIP(hipDeviceSynchronize());
    DH_HIP(hipMemcpy(out64_host, g_stamps_dev, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    DH_HIP(hipMemset(g_stamps_dev, 0, 64 * sizeof(unsigned long long)));
  }
  return DH_OK;
}

#include "gemm1x1_f32.inc"
#include "train.inc"
#include "train2_kernels.inc"
#include "wgrad_ring.inc"
#include "bn_fold.inc"
#include "gemm1x1.inc"
#include "train2.inc"
