// Sanitizer sweep of the HOST tables behind conv3x3_kernel (deephisto_amd/csrc/conv3_tables_host.h) -- VERDICT r3 item 6.
// Built by tests/test_conv_tables_host.py with `g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all`.
//
// For every (dtype, layout, patch P, tiles per launch n, layer of ResNet-18 / the 3x3 layers of ResNet-50, stride, tile
// candidate) it builds the lane / tile / mask tables exactly as the library does and replays the kernel's address arithmetic
// on the host (conv3x3.inc: enter_tile, dma_piece, epilogue_of):
//   * every window piece whose mask bit is set reads 16 bytes INSIDE the input tensor, at the pixel / channel slot the LDS image
//     expects, and every in-image pixel of the staged window IS covered by a set bit (no padding where data should be);
//   * every output pixel x cout block is written exactly once, inside the output tensor;
//   * window byte offsets and output element offsets fit the 32-bit unsigned words of the schedule, per-lane offsets fit int32;
//   * a workgroup keeps one cout block over all iterations (resident weights), mask rows < 4096.
// Prints one line per case class and "OK <cases>"; any violation aborts with a message (non-zero exit).
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <string>

#include "../../deephisto_amd/csrc/conv3_tables_host.h"

using namespace dh_conv3;

struct Geom {   // the geometry fields of Conv3Params
  int B, Hi, Wi, Cin, Ho, Wo, Cout;
  int in_px_bytes, in_chunk_bytes, out_px, out_cb, out_mt;
  int64_t o_img; int o_row, o_px, o_base;
  int out_pr, res_mt, res_pr, r_row, r_px, r_cb, r_base;   // output half-tile stride; the residual's own layout
  int TH, TW, IMGS, tiles_y, tiles_x, HR, HC, HP, HPH, WTAIL, FIT, IP, n_win_instr, ntiles, iters;
};

static long g_cases = 0;
#define REQUIRE(cond, ...) do { if (!(cond)) { fprintf(stderr, "VIOLATION %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); abort(); } } while (0)

// layout of the output / input: 0 NHWC, 1 channel-blocked in 32-channel planes, 2 in 16-channel planes (out16 / in16)
template <int STRIDE, int NT, int WAVES, int ESZ, int MT, bool HALF = false>
static void check_case(Geom p, int in_layout, int out_layout, int grid_override, const char* what) {
  constexpr int MAXJ = max_window_pieces<STRIDE, NT, WAVES>();
  constexpr int CO_BLK = HALF ? 128 : 64, PXB = HALF ? 32 : 64;   // couts per workgroup; bytes of a pixel per stage
  const bool blocked = in_layout != 0;
  const int ncb = p.Cout / CO_BLK;
  const int win_px = p.IMGS * p.HR * p.HP + p.WTAIL;
  if (!p.IP) p.IP = p.HR * p.HP;
  p.n_win_instr = HALF ? (win_px + 31) / 32 : (win_px + 15) / 16;
  REQUIRE(p.n_win_instr <= MAXJ * WAVES, "%s: window too large for the DMA plan", what);
  const int slots = (HALF ? WAVES / 2 : WAVES * MT / 2) * NT * 32;
  REQUIRE(p.FIT ? p.IMGS * p.TH * p.TW <= slots : p.IMGS * p.TH * p.TW == slots, "%s: tile/pixel mismatch", what);
  // the LDS budget of the launch (launch_conv3x3_cfg): two ring slots of [36 KiB weight slab (+ 4 KiB downsample slab) | window] + tables
  REQUIRE(2 * ((size_t)(9 + (STRIDE == 2 && !HALF ? 1 : 0)) * 4096 + (((size_t)win_px * PXB + 1023) & ~(size_t)1023)) + (HALF ? 4096 : 2048) <= 160 * 1024 || HALF,
          "%s: window of %d pixels does not fit the LDS ring", what, win_px);
  const int groups = ((p.B + p.IMGS - 1) / p.IMGS) * p.tiles_y * p.tiles_x;
  p.ntiles = groups * ncb;
  const int grid = grid_override > 0 ? grid_override : std::min(256, p.ntiles);
  p.iters = (p.ntiles + grid - 1) / grid;
  HostTables ht;
  const char* why = build_tables<STRIDE, NT, WAVES, ESZ, MT, Geom, HALF>(p, ncb, grid_override, &ht);
  const int64_t in_bytes = (int64_t)p.B * p.Hi * p.Wi * p.Cin * ESZ;
  const int64_t out_elems = (int64_t)p.B * p.o_img;
  if (why) {   // the only legitimate refusal: a tensor past 4 Gi bytes / elements
    REQUIRE(in_bytes >= ((int64_t)1 << 32) || out_elems >= ((int64_t)1 << 32), "%s: refused (%s) although the tensors fit", what, why);
    ++g_cases;
    return;
  }
  REQUIRE(ht.grid == grid && ht.mask_rows < 4096 && (int)ht.tile.size() == p.iters * grid, "%s: table sizes", what);
  const int threads = WAVES * 64;
  const int nchunks = p.Cin * ESZ / PXB;
  // written[(pixel, cout block)] exactly once
  std::vector<uint8_t> written((size_t)p.B * p.Ho * p.Wo * ncb, 0);
  std::vector<int> wg_cb(grid, -1);
  std::vector<uint8_t> staged;
  for (int it = 0; it < p.iters; ++it)
    for (int w = 0; w < grid; ++w) {
      const TileDesc td = ht.tile[(size_t)it * grid + w];
      const int cb = td.x & 0xFFFF, valid = (td.x >> 16) & 1, mrow = (unsigned)td.x >> 20;
      REQUIRE(cb < ncb && mrow < ht.mask_rows, "%s: descriptor fields", what);
      if (wg_cb[w] < 0) wg_cb[w] = cb;
      REQUIRE(wg_cb[w] == cb, "%s: workgroup %d changes its cout block (%d -> %d)", what, w, wg_cb[w], cb);
      if (!valid) continue;
      const uint32_t win_off = (uint32_t)td.y, out_off = (uint32_t)td.z;
      // recover the tile position from the window offset (NHWC and blocked alike: img * img_bytes + (y * Wi + x) * px_bytes)
      const int64_t img_bytes = (int64_t)p.Hi * p.Wi * p.Cin * ESZ;
      const int img0 = (int)(win_off / img_bytes);
      const int64_t rpx = (win_off % img_bytes) / p.in_px_bytes;
      REQUIRE((win_off % img_bytes) % p.in_px_bytes == 0, "%s: window offset not on a pixel", what);
      const int iy0 = (int)(rpx / p.Wi), ix0 = (int)(rpx % p.Wi);
      REQUIRE(iy0 % STRIDE == 0 && ix0 % STRIDE == 0, "%s: window origin off the stride grid", what);
      const int oy0 = iy0 / STRIDE, ox0 = ix0 / STRIDE;
      const int imgs_here = std::min(p.IMGS, p.B - img0);
      REQUIRE(imgs_here > 0, "%s: tile beyond the batch", what);
      staged.assign((size_t)win_px * 4, 0);   // [LDS pixel][16-byte slot]: filled from the tensor?
      for (int tid = 0; tid < threads; ++tid) {
        const int* row = &ht.lane[(size_t)tid * ht.lane_stride];
        const unsigned mk = ht.mask[(size_t)mrow * threads + tid];
        const int l = tid & 63, wave = tid >> 6;
        // ---- window pieces (dma_piece): src = in + win_off + chunk * in_chunk_bytes + rel_off[j], 16 bytes
        for (int j = 0; j < MAXJ; ++j) {
          const int i = wave + WAVES * j;
          if (i >= p.n_win_instr) { REQUIRE(!((mk >> j) & 1u), "%s: mask bit on an unissued piece", what); continue; }
          const int px = HALF ? i * 32 + (l >> 1) : i * 16 + (l >> 2);
          if (!((mk >> j) & 1u)) continue;
          for (int ch = 0; ch < nchunks; ch += std::max(1, nchunks - 1)) {   // first and last chunk
            const int64_t src = (int64_t)win_off + (int64_t)ch * p.in_chunk_bytes + row[3 * NT + j];
            REQUIRE(src >= 0 && src + 16 <= in_bytes, "%s: window piece reads [%" PRId64 ", +16) outside the %" PRId64 "-byte input (tile it %d wg %d tid %d j %d)",
                    what, src, in_bytes, it, w, tid, j);
            // which pixel / slot is it?  NHWC: byte = ((img * Hi + y) * Wi + x) * Cin * ESZ + chunk * 64 + slot * 16
            //                            blocked: byte = (img * nchunks + chunk) * Hi * Wi * 64 + (y * Wi + x) * 64 + slot * 16
            int64_t simg, sy, sx, sch, sslot;
            if (in_layout == 2) {          // 16-channel planes: [image][C/16][H][W][16] -- a plane IS a half-chunk
              simg = src / img_bytes; const int64_t r = src % img_bytes;
              sch = r / ((int64_t)p.Hi * p.Wi * 32); const int64_t r2 = r % ((int64_t)p.Hi * p.Wi * 32);
              sy = r2 / (p.Wi * 32); sx = (r2 % (p.Wi * 32)) / 32; sslot = (r2 % 32) / 16;
            } else if (blocked) {
              simg = src / img_bytes; const int64_t r = src % img_bytes;
              sch = r / ((int64_t)p.Hi * p.Wi * 64); const int64_t r2 = r % ((int64_t)p.Hi * p.Wi * 64);
              sy = r2 / (p.Wi * 64); sx = (r2 % (p.Wi * 64)) / 64; sslot = (r2 % 64) / 16;
            } else {
              simg = src / img_bytes; const int64_t r = src % img_bytes;
              const int64_t pxb = (int64_t)p.Cin * ESZ;
              sy = r / (p.Wi * pxb); sx = (r % (p.Wi * pxb)) / pxb; sch = ((r % pxb) / 64); sslot = (r % 64) / 16;
            }
            const int limg = px / p.IP, lr = px % p.IP;
            REQUIRE(lr < p.HR * p.HP, "%s: a gap pixel of the window is staged", what);
            const int hy = lr / p.HP, c = lr % p.HP;
            const int hx = STRIDE == 2 ? 2 * (c % p.HPH) + c / p.HPH : c;
            REQUIRE(simg == img0 + limg && sy == iy0 + hy - 1 && sx == ix0 + hx - 1 && sch == ch, "%s: window piece lands on the wrong pixel", what);
            if (HALF) { REQUIRE(sslot == ((l & 1) ^ ((px >> 3) & 1)), "%s: swizzle slot (half)", what); if (ch == 0) { staged[(size_t)px * 4 + (l & 1)] = 1; staged[(size_t)px * 4 + 2 + (l & 1)] = 1; } }
            else { REQUIRE(sslot == ((l & 3) ^ ((px >> 2) & 3)), "%s: swizzle slot", what); if (ch == 0) staged[(size_t)px * 4 + (l & 3)] = 1; }
          }
        }
        // ---- output pixels (epilogue_of): out + out_off + out_rel[nt] + mt * out_mt + 32 couts
        for (int nt = 0; nt < NT; ++nt) {
          if (!((mk >> (16 + nt)) & 1u)) continue;
          const int64_t base = (int64_t)out_off + row[nt];
          REQUIRE(base <= UINT32_MAX, "%s: output offset past 32 bits", what);
          for (int ml = 0; ml < MT; ++ml) {
            const int mt = HALF ? 2 * (wave & 1) + ml : MT == 2 ? ml : (wave & 1);
            for (int pr = 0; pr < 2; ++pr) {   // the two 16-cout halves of a 32-cout tile (epilogue_of: + pr * out_pr + h * 8)
              const int64_t e0 = base + (int64_t)mt * p.out_mt + (int64_t)pr * p.out_pr;
              REQUIRE(e0 >= 0 && e0 + 16 <= out_elems, "%s: output [%" PRId64 ", +16) outside %" PRId64 " elements", what, e0, out_elems);
            }
          }
          {   // the residual tile (prefetch_residual: res + td.w + res_rel + mt * res_mt + pr * res_pr + h * 8), always in its own layout
            const int64_t rbase = (int64_t)(uint32_t)td.w + row[2 * NT + nt];
            for (int ml = 0; ml < MT; ++ml) {
              const int mt = HALF ? 2 * (wave & 1) + ml : MT == 2 ? ml : (wave & 1);
              const int64_t e0 = rbase + (int64_t)mt * p.res_mt + p.res_pr;
              REQUIRE(e0 >= 0 && e0 + 16 <= out_elems, "%s: residual [%" PRId64 ", +16) outside %" PRId64 " elements", what, e0, out_elems);
            }
            // ... and it is the SAME pixel / cout block as the output (NHWC / 32-channel planes)
            int64_t rpix;
            if (blocked) { const int64_t r = rbase % p.o_img; rpix = (rbase / p.o_img) * p.Ho * p.Wo + (r % ((int64_t)p.Ho * p.Wo * 32)) / 32; REQUIRE(r / ((int64_t)p.Ho * p.Wo * 32) == (CO_BLK / 32) * cb, "%s: residual cout block", what); }
            else { rpix = rbase / p.Cout; REQUIRE(rbase % p.Cout == cb * CO_BLK, "%s: residual cout block (NHWC)", what); }
            int64_t opix;
            if (out_layout == 2) { const int64_t r = base % p.o_img; opix = (base / p.o_img) * p.Ho * p.Wo + (r % ((int64_t)p.Ho * p.Wo * 16)) / 16; }
            else if (out_layout == 1) { const int64_t r = base % p.o_img; opix = (base / p.o_img) * p.Ho * p.Wo + (r % ((int64_t)p.Ho * p.Wo * 32)) / 32; }
            else opix = base / p.Cout;
            REQUIRE(rpix == opix, "%s: residual pixel %" PRId64 " != output pixel %" PRId64, what, rpix, opix);
          }
          if ((l >> 5) == 0 && ((MT == 2 && !HALF) || (wave & 1) == 0)) {   // one owner per pixel: half-wave 0 (the other half holds the other couts)
            // pixel index from the element offset (dense maps only: o_px == out_px)
            if (p.o_px == p.out_px && p.o_base == 0) {
              int64_t pix;
              if (out_layout == 2) { const int64_t r = base % p.o_img; pix = (base / p.o_img) * p.Ho * p.Wo + (r % ((int64_t)p.Ho * p.Wo * 16)) / 16; REQUIRE(r / ((int64_t)p.Ho * p.Wo * 16) == (CO_BLK / 16) * cb, "%s: 16-plane cout block", what); }
              else if (out_layout == 1) { const int64_t r = base % p.o_img; pix = (base / p.o_img) * p.Ho * p.Wo + (r % ((int64_t)p.Ho * p.Wo * 32)) / 32; REQUIRE(r / ((int64_t)p.Ho * p.Wo * 32) == (CO_BLK / 32) * cb, "%s: blocked cout block", what); }
              else { pix = base / p.Cout; REQUIRE(base % p.Cout == cb * CO_BLK, "%s: NHWC cout block", what); }
              REQUIRE(pix >= 0 && pix < (int64_t)p.B * p.Ho * p.Wo, "%s: pixel index", what);
              uint8_t& f = written[(size_t)pix * ncb + cb];
              REQUIRE(f == 0, "%s: output pixel %" PRId64 " cout block %d written twice", what, pix, cb);
              f = 1;
            }
          }
        }
      }
      // every tap of every existing output pixel reads INSIDE the window, on a pixel staged from the tensor exactly when the tap is inside
      // the image and on a zero (unstaged) pixel otherwise -- this is what makes shared halo rows / columns of fit tiles legal
      if (STRIDE == 1 && !HALF)
        for (int tid = 0; tid < threads; tid += 1) {
          if ((tid & 63) >= 32) continue;   // lanes l and l + 32 own the same pixel
          const int* row = &ht.lane[(size_t)tid * ht.lane_stride];
          const unsigned mk = ht.mask[(size_t)mrow * threads + tid];
          for (int nt = 0; nt < NT; ++nt) {
            if (!((mk >> (16 + nt)) & 1u)) continue;
            const int base_lin = row[NT + nt];
            // recover (img, ty, tx) of the pixel from its output offset (dense NHWC / blocked maps only)
            if (!(p.o_px == p.out_px && p.o_base == 0)) continue;
            const int64_t e = (int64_t)out_off + row[nt];
            int64_t pix;
            if (out_layout == 2) pix = (e / p.o_img) * p.Ho * p.Wo + ((e % p.o_img) % ((int64_t)p.Ho * p.Wo * 16)) / 16;
            else if (out_layout == 1) pix = (e / p.o_img) * p.Ho * p.Wo + ((e % p.o_img) % ((int64_t)p.Ho * p.Wo * 32)) / 32;
            else pix = e / p.Cout;
            const int oy = (int)((pix % ((int64_t)p.Ho * p.Wo)) / p.Wo), ox = (int)(pix % p.Wo);
            for (int kh = 0; kh < 3; ++kh)
              for (int kw = 0; kw < 3; ++kw) {
                const int lin = base_lin + kh * p.HP + kw;
                REQUIRE(lin >= 0 && lin < win_px, "%s: tap (%d, %d) of pixel (%d, %d) reads window index %d of %d", what, kh, kw, oy, ox, lin, win_px);
                const int iy = oy + kh - 1, ix = ox + kw - 1;
                const bool inside = iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
                REQUIRE((staged[(size_t)lin * 4] != 0) == inside, "%s: tap (%d, %d) of pixel (%d, %d): window index %d is %s but the tap is %s the image",
                        what, kh, kw, oy, ox, lin, staged[(size_t)lin * 4] ? "data" : "zero", inside ? "inside" : "outside");
              }
          }
        }
      // every in-image pixel of the window must have all four slots staged from the tensor
      for (int limg = 0; limg < imgs_here; ++limg)
        for (int hy = 0; hy < p.HR; ++hy)
          for (int hx = 0; hx < p.HC; ++hx) {
            const int iy = iy0 + hy - 1, ix = ix0 + hx - 1;
            if (iy < 0 || iy >= p.Hi || ix < 0 || ix >= p.Wi) continue;
            const int c = STRIDE == 2 ? (hx & 1) * p.HPH + (hx >> 1) : hx;
            const size_t px = (size_t)limg * p.IP + (size_t)hy * p.HP + c;
            for (int sl = 0; sl < 4; ++sl)
              REQUIRE(staged[px * 4 + sl], "%s: in-image window pixel (%d, %d, %d) slot %d is padded", what, limg, hy, hx, sl);
          }
    }
  if (p.o_px == p.out_px && p.o_base == 0)
    for (size_t i = 0; i < written.size(); ++i) REQUIRE(written[i], "%s: output pixel %zu / cout block %zu never written", what, i / ncb, i % ncb);
  ++g_cases;
}

// out16: the output in 16-channel planes (the conv before a wide stride-2 conv); wide: the wide stride-2 variant itself (input in 16-channel planes)
template <int ESZ>
static void sweep_layer(int B, int Hi, int cin, int cout, int stride, bool blocked, int min_tiles, const std::string& tag, bool out16 = false, bool wide = false) {
  Geom p{};
  p.B = B; p.Hi = Hi; p.Wi = Hi; p.Cin = cin; p.Cout = cout;
  p.Ho = (Hi + 2 - 3) / stride + 1; p.Wo = p.Ho;
  if (blocked) { p.in_px_bytes = kChunkBytes; p.in_chunk_bytes = Hi * Hi * kChunkBytes; p.out_px = 32; p.out_mt = p.Ho * p.Wo * 32; p.out_cb = 2 * p.out_mt; }
  else { p.in_px_bytes = cin * ESZ; p.in_chunk_bytes = kChunkBytes; p.out_px = cout; p.out_mt = 32; p.out_cb = 64; }
  p.o_img = (int64_t)p.Ho * p.Wo * cout; p.o_row = p.Wo * p.out_px; p.o_px = p.out_px; p.o_base = 0;
  p.out_pr = 16; p.res_mt = p.out_mt; p.res_pr = 16; p.r_row = p.o_row; p.r_px = p.o_px; p.r_cb = p.out_cb; p.r_base = 0;
  if (out16) { p.out_px = 16; p.out_pr = p.Ho * p.Wo * 16; p.out_mt = 2 * p.out_pr; p.out_cb = 2 * p.out_mt; p.o_row = p.Wo * 16; p.o_px = 16; }
  if (wide) { p.in_px_bytes = 32; p.in_chunk_bytes = Hi * Hi * 32; }
  const std::string what = tag + " B=" + std::to_string(B) + " Hi=" + std::to_string(Hi) + " " + std::to_string(cin) + "->" + std::to_string(cout) +
                           " s" + std::to_string(stride) + (blocked ? " blocked" : " nhwc") + (out16 ? " out16" : "") + (wide ? " wide" : "") + " esz" + std::to_string(ESZ);
  const int in_l = wide ? 2 : blocked ? 1 : 0, out_l = out16 ? 2 : blocked ? 1 : 0;
  if (stride == 2) {
    if (wide) {
      if constexpr (ESZ == 2) {
        REQUIRE(set_stride2_wide_geometry(p, p.Ho, p.Wo), "%s: no wide geometry", what.c_str());
        p.out_cb = 4 * p.out_mt; p.r_cb = p.out_cb;
        check_case<2, 2, 8, ESZ, 2, true>(p, in_l, out_l, 0, what.c_str());
      }
      return;
    }
    set_stride2_geometry(p, p.Ho, p.Wo);
    check_case<2, 1, 8, ESZ, 1>(p, in_l, out_l, 0, what.c_str());
    return;
  }
  Cand c[kMaxCands];
  const int nc = stride1_candidates(p.Ho, p.Wo, c);
  const Cand picked = pick_stride1(B, p.Ho, p.Wo, cout, min_tiles);
  for (int i = 0; i < nc; ++i) {   // ALL candidates, not only the picked one
    set_stride1_geometry(p, c[i], p.Ho, p.Wo);
    const std::string w2 = what + " cand" + std::to_string(i) + (c[i].fit ? "fit" : "") + (c[i].th == picked.th && c[i].tw == picked.tw && c[i].imgs == picked.imgs ? "*" : "");
    if (c[i].variant == 0) check_case<1, 2, 8, ESZ, 2>(p, in_l, out_l, 0, w2.c_str());
    else if (c[i].variant == 1) check_case<1, 1, 8, ESZ, 2>(p, in_l, out_l, 0, w2.c_str());
    else check_case<1, 1, 8, ESZ, 1>(p, in_l, out_l, 0, w2.c_str());
  }
}

int main(int argc, char** argv) {
  const bool full = argc > 1 && std::string(argv[1]) == "full";
  const int Ps[] = {32, 64, 96, 100, 224, 256, 330};
  const int ns_small[] = {1, 3, 64};
  const int ns_big[] = {1024, 3842, 4096};
  // ResNet-18 body: (cin, cout, stride, input map = pooled size / 2^stage)
  struct L { int cin, cout, stride, shift; };
  const L layers[] = {{64, 64, 1, 0}, {64, 128, 2, 0}, {128, 128, 1, 1}, {128, 256, 2, 1}, {256, 256, 1, 2}, {256, 512, 2, 2}, {512, 512, 1, 3}};
  for (int P : Ps) {
    const int H1 = (P + 6 - 7) / 2 + 1, H2 = (H1 + 2 - 3) / 2 + 1;
    for (const L& ly : layers) {
      int Hi = H2;
      for (int s = 0; s < ly.shift; ++s) Hi = (Hi + 2 - 3) / 2 + 1;
      for (int n : ns_small) {
        sweep_layer<2>(n, Hi, ly.cin, ly.cout, ly.stride, true, 256, "inf-bf16");    // bf16 inference: channel-blocked
        if (ly.stride == 1 && ly.cin == ly.cout) sweep_layer<2>(n, Hi, ly.cin, ly.cout, 1, true, 256, "inf-bf16", true);   // ... writing 16-channel planes
        if (ly.stride == 2 && ly.cout % 128 == 0 && (Hi + 2 - 3) / 2 + 1 > 4) sweep_layer<2>(n, Hi, ly.cin, ly.cout, 2, true, 256, "inf-bf16", false, true);   // the wide stride-2 variant
        sweep_layer<2>(n, Hi, ly.cin, ly.cout, ly.stride, false, 256, "train-bf16"); // bf16 training: NHWC
        sweep_layer<4>(n, Hi, ly.cin, ly.cout, ly.stride, false, 256, "f32");
      }
      // the launch sizes the headline number is timed at: only where the coverage bitmap stays small enough for a unit test,
      // i.e. the large n on the layers of P in {64, 256} (P = 256, n = 4096, layer 1 is THE 2^31-byte map of VERDICT r3 missing #1)
      if (P == 256 || P == 64 || full)
        for (int n : ns_big) {
          if (!full && P == 256 && ly.shift > 1 && n != 4096) continue;
          sweep_layer<2>(n, Hi, ly.cin, ly.cout, ly.stride, true, 256, "inf-bf16");
          if (ly.stride == 1 && ly.cin == ly.cout && ly.shift < 3) sweep_layer<2>(n, Hi, ly.cin, ly.cout, 1, true, 256, "inf-bf16", true);
          if (ly.stride == 2 && ly.cout % 128 == 0 && (Hi + 2 - 3) / 2 + 1 > 4) sweep_layer<2>(n, Hi, ly.cin, ly.cout, 2, true, 256, "inf-bf16", false, true);
          if (n <= 1024) sweep_layer<4>(n, Hi, ly.cin, ly.cout, ly.stride, false, 256, "f32");
        }
    }
    printf("P=%d done (%ld cases)\n", P, g_cases);
    fflush(stdout);
  }
  // float32 at 4 096 tiles of 256^2: the layer-1 map is 2^32 bytes -- must be REFUSED, not wrapped
  sweep_layer<4>(4096, 64, 64, 64, 1, false, 256, "f32-oversize");
  // odd maps of training shapes (224 -> 56 / 28 / 14 / 7) at batch 64 with the 3x3 layers of ResNet-50
  for (int Hi : {56, 28, 14, 7})
    for (int c : {64, 128, 256, 512}) {
      sweep_layer<2>(64, Hi, c, c, 1, false, 256, "r50-3x3");
      if (Hi > 7) sweep_layer<2>(64, Hi, c, c, 2, false, 256, "r50-3x3");
    }
  // ---- refusal paths: the invariants the kernel relies on are refused by build_tables, not assumed by its callers
  {
    auto geom = [](int B, int Hi, int Wi, int cin, int cout) {
      Geom p{};
      p.B = B; p.Hi = Hi; p.Wi = Wi; p.Cin = cin; p.Cout = cout; p.Ho = Hi; p.Wo = Wi;
      p.in_px_bytes = cin * 2; p.in_chunk_bytes = kChunkBytes; p.out_px = cout; p.out_mt = 32; p.out_cb = 64;
      p.o_img = (int64_t)p.Ho * p.Wo * cout; p.o_row = p.Wo * p.out_px; p.o_px = p.out_px; p.o_base = 0;
      p.out_pr = 16; p.res_mt = p.out_mt; p.res_pr = 16; p.r_row = p.o_row; p.r_px = p.o_px; p.r_cb = p.out_cb; p.r_base = 0;
      return p;
    };
    auto refused = [](Geom p, const Cand& c, const char* expect) {
      set_stride1_geometry(p, c, p.Ho, p.Wo);
      p.n_win_instr = (p.IMGS * p.HR * p.HP + 15) / 16;
      const int ncb = p.Cout / 64;
      p.ntiles = ((p.B + p.IMGS - 1) / p.IMGS) * p.tiles_y * p.tiles_x * ncb;
      const int grid = std::min(256, p.ntiles);
      p.iters = (p.ntiles + grid - 1) / grid;
      HostTables ht;
      const char* why = build_tables<1, 2, 8, 2, 2, Geom>(p, ncb, 0, &ht);
      REQUIRE(why && std::string(why).find(expect) != std::string::npos, "expected a refusal containing '%s', got '%s'", expect, why ? why : "(accepted)");
      ++g_cases;
    };
    const Cand c512 = {16, 32, 1, 34, 0};
    refused(geom(4, 64, 64, 32, 64), c512, "two whole channel stages");          // bf16, 32 channels: ONE 64-byte stage per tile
    refused(geom(4, 64, 64, 80, 64), c512, "two whole channel stages");          // 160 bytes per pixel: not whole stages
    refused(geom(1, 1040, 2048, 64, 64), c512, "too many distinct tile positions");   // 65 x 64 = 4 160 tile positions > 4 096 mask rows
    {   // exactly 4 096 positions (mask rows 0 .. 4095: the descriptor word's bit 31 is used) is accepted and decodes
      Geom p = geom(1, 1024, 2048, 64, 64);
      set_stride1_geometry(p, c512, p.Ho, p.Wo);
      p.n_win_instr = (p.IMGS * p.HR * p.HP + 15) / 16;
      p.ntiles = p.tiles_y * p.tiles_x; p.iters = (p.ntiles + 255) / 256;
      HostTables ht;
      const char* why = build_tables<1, 2, 8, 2, 2, Geom>(p, 1, 0, &ht);
      REQUIRE(!why && ht.mask_rows == 4096, "4 096 tile positions: %s, %d rows", why ? why : "accepted", ht.mask_rows);
      unsigned top = 0;
      for (const TileDesc& td : ht.tile) top = std::max(top, (unsigned)td.x >> 20);
      REQUIRE(top == 4095, "highest mask row decodes as %u", top);
      ++g_cases;
    }
    {   // a window past 64 KiB: 24 x 44 pixels x 64 bytes = 66 KiB (the 16-bit hoisted offsets would wrap)
      Geom p = geom(4, 64, 64, 64, 64);
      Cand big = {22, 32, 1, 44, 0};
      set_stride1_geometry(p, big, p.Ho, p.Wo);
      p.n_win_instr = (p.IMGS * p.HR * p.HP + 15) / 16;
      p.ntiles = 4 * p.tiles_y * p.tiles_x; p.iters = 1;
      HostTables ht;
      const char* why = build_tables<1, 2, 8, 2, 2, Geom>(p, 1, 0, &ht);
      REQUIRE(why && std::string(why).find("64 KiB") != std::string::npos, "oversized window: %s", why ? why : "(accepted)");
      ++g_cases;
    }
  }
  // ---- fit tiles: bank-conflict census of the dealt lane order (a 16-lane service group reads conflict-free when its pixels have 16
  // different window indices mod 16)
  {
    struct F { int imgs, th, tw, ip, hp, ntile32, max_conflict_groups; const char* name; };
    for (const F& f : {F{10, 7, 7, 67, 8, 16, 0, "7x7 x10"}, F{5, 7, 7, 67, 8, 8, 8, "7x7 x5"}, F{5, 7, 14, 135, 15, 16, 0, "7x14 x5"}, F{8, 8, 8, 82, 9, 16, 0, "8x8 x8"}}) {
      const std::vector<int> slot = deal_fit_pixels(f.imgs, f.th, f.tw, f.ip, f.hp, f.ntile32);
      std::set<int> seen;
      int conflict_groups = 0;
      static const int ga[16] = {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27};
      for (int t = 0; t < f.ntile32; ++t)
        for (int half = 0; half < 2; ++half) {
          int cnt[16] = {0};
          bool in_a[32] = {false};
          for (int k = 0; k < 16; ++k) in_a[ga[k]] = true;
          for (int ll = 0; ll < 32; ++ll) {
            if (in_a[ll] != (half == 0)) continue;
            const int q = slot[(size_t)t * 32 + ll];
            if (q < 0) continue;
            REQUIRE(seen.insert(q).second, "%s: pixel %d dealt twice", f.name, q);
            const int img = q / (f.th * f.tw), ty = (q % (f.th * f.tw)) / f.tw, tx = q % f.tw;
            ++cnt[(img * f.ip + ty * f.hp + tx) & 15];
          }
          bool c2 = false;
          for (int r = 0; r < 16; ++r) { REQUIRE(cnt[r] <= 2, "%s: %d-way conflict", f.name, cnt[r]); c2 |= cnt[r] > 1; }
          conflict_groups += c2;
        }
      REQUIRE((int)seen.size() == f.imgs * f.th * f.tw, "%s: %zu of %d pixels dealt", f.name, seen.size(), f.imgs * f.th * f.tw);
      REQUIRE(conflict_groups <= f.max_conflict_groups, "%s: %d service groups with a 2-way conflict (bound %d)", f.name, conflict_groups, f.max_conflict_groups);
      printf("fit %s: %d of %d service groups with a 2-way bank conflict\n", f.name, conflict_groups, 2 * f.ntile32);
      ++g_cases;
    }
  }
  printf("OK %ld\n", g_cases);
  return 0;
}
