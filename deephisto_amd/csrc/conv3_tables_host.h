// conv3_tables_host.h -- HOST-ONLY integer code behind the 3x3 convolution kernel (conv3x3.inc): tile-shape choice, per-thread
// geometry, per-tile schedule and mask tables.  Every LDS-DMA source address and every output address of the kernel comes out of
// these tables, so the file is kept free of HIP: resnet_kernels.hip includes it for the product, and
// tests/host/conv3_tables_sweep.cpp compiles it with `g++ -fsanitize=address,undefined` and sweeps the shapes the engines use,
// asserting that every offset lies inside its tensor and fits its integer type (tests/test_conv_tables_host.py, CPU only).
//
// `P` is any struct with the geometry fields of Conv3Params (TH, TW, IMGS, HR, HC, HP, HPH, WTAIL, FIT, IP, Hi, Wi, Cin, B, tiles_y, tiles_x,
// n_win_instr, in_px_bytes, Ho, Wo, Cout, out_px, out_cb, o_img, o_row, o_px, o_base, r_row, r_px, r_cb, r_base, ntiles, iters).
//
// Round 5: "fit" tile shapes for maps whose side is 7 * 2^k (every layer of a 224-pixel patch: 56 / 28 / 14 / 7).  Power-of-two tiles
// fill 49 / 64 of their slots there.  A fit tile takes whole 7 x 7 images (ten per 512-slot tile, 490 slots) or half-images of 7 x 14
// (five per tile, 490 slots); the slot count need not be filled (FIT: IMGS * TH * TW <= slots, the rest are masked lanes), and where a
// tile covers the map's full width / height the zero halo column / row is SHARED between neighbouring rows / images (HP = TW + 1,
// HR = TH + 1): the pixel right of a row's last pixel is the (always zero) left halo of the next row, so the staged window is
// IMGS * HR * HP + WTAIL pixels instead of IMGS * (TH + 2) * (TW + 2) -- that is what makes ten 7 x 7 images (649 pixels) and five
// 7 x 14 half-images (676) fit the 43 KiB a window may take beside two 36 KiB weight slabs.  Lanes are dealt to pixels by the
// residue of their window index mod 16 (deal_fit_pixels), the condition under which the swizzled window image is read conflict-free;
// the image segments of a fit window are IP pixels apart (IP >= HR * HP: the gap pixels are never staged, i.e. zero), chosen so that
// the residues come out even: at pitch 8 the 7-pixel rows of ten images at IP = 64 never produce residues 7 and 15 (21 of 32 service
// groups would read with 2-way conflicts), at IP = 67 every residue occurs at most 32 times: conflict-free.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <map>
#include <vector>

namespace dh_conv3 {

constexpr int kChunkBytes = 64;   // channel bytes of one pixel staged per pass (CHUNK_BYTES)

struct TileDesc { int x, y, z, w; };   // {cout block | valid << 16 | mask row << 20, window byte offset, output element offset, residual element offset}

struct HostTables {
  std::vector<int> lane;        // [threads][3 NT + MAXJ] = {out_rel[NT], base_lin[NT], res_rel[NT], rel_off[MAXJ]}
  std::vector<TileDesc> tile;   // [iters][grid]
  std::vector<unsigned> mask;   // [mask rows][threads]: bits 0..MAXJ-1 window piece j inside the image, bit 16 + nt: pixel nt exists
  int grid = 0, threads = 0, lane_stride = 0, mask_rows = 0;
  bool xcd_group = false;
};

template <int STRIDE, int NT, int WAVES>
constexpr int max_window_pieces() { return (STRIDE == 2) ? (WAVES == 8 ? 5 : 10) : (NT == 2 ? 6 : 4); }

// Tile shapes of the stride-1 kernel from the largest down: {TH, TW, IMGS, HP, variant}; variant 0: NT=2 MT=2 (512 pixels),
// 1: NT=1 MT=2 (256), 2: NT=1 MT=1 (128).  The first candidate that gives the launch `min_tiles` tiles wins, else the smallest.
// hr = 0: TH + 2 window rows per image segment (own halo rows); wtail: window pixels behind the last segment; fit: see the header
struct Cand { int th, tw, imgs, hp, variant, hr = 0, wtail = 0, fit = 0, ip = 0; };   // ip = 0: segments HR * HP pixels apart
constexpr int kMaxCands = 6;
inline int stride1_candidates(int Ho, int Wo, Cand (&c)[kMaxCands], bool fit = true) {
  int nc = 0;
  if (fit && Ho == 7 && Wo == 7) {
    // whole 7 x 7 images, halo row and column shared (pitch 8, 8 rows per image; tail = one halo row + 1): 10 images = 490 / 512 slots,
    // 5 images = 245 / 256
    // segments 67 pixels apart (see the header); window = (IMGS - 1) * 67 + 8 * 8 + 9 pixels = IMGS * 64 + wtail
    c[nc++] = {7, 7, 10, 8, 0, 8, 9 * 3 + 9, 1, 67};
    c[nc++] = {7, 7, 5, 8, 1, 8, 4 * 3 + 9, 1, 67};
    c[nc++] = {8, 8, 2, 12, 2};
    return nc;
  }
  if (fit && Ho == 14 && Wo == 14) {
    // half-images of 7 rows x 14 columns from five images: 490 / 512 slots, 2 tiles per 5 images; the halo COLUMN is shared (full
    // width: pitch 15), the halo rows are real rows of the image (HR = 9); tail = the last row's right halo
    c[nc++] = {7, 14, 5, 15, 0, 9, 1, 1};
    c[nc++] = {16, 16, 1, 18, 1};
    c[nc++] = {8, 8, 2, 12, 2};
    return nc;
  }
  if (fit && Ho == 8 && Wo == 8) {
    // whole 8 x 8 images with shared zero halos (pitch 9, 9 rows per image), segments 82 pixels apart (every residue mod 16 exactly 32
    // times: conflict-free): EIGHT images = 512 slots on the NT = 2 variant -- the power-of-two list runs 8 x 8 maps as four images
    // per 256-slot tile (NT = 1: a 36 KiB weight slab per 256 pixels instead of per 512; layer 4 of a 256-pixel patch, cin = 512)
    c[nc++] = {8, 8, 8, 9, 0, 9, 7 * 82 + 81 + 10 - 8 * 81, 1, 82};
    c[nc++] = {8, 8, 4, 12, 1};
    c[nc++] = {8, 8, 2, 12, 2};
    return nc;
  }
  if (Wo > 16) {
    // 16x32 or 8x64 output pixels, whichever wastes fewer tile slots (56x56: 77 % vs 88 % useful)
    const int slots_a = ((Ho + 15) / 16) * ((Wo + 31) / 32), slots_b = ((Ho + 7) / 8) * ((Wo + 63) / 64);
    if (slots_b < slots_a) c[nc++] = {8, 64, 1, 66, 0};
    else c[nc++] = {16, 32, 1, 34, 0};
    c[nc++] = {16, 16, 1, 18, 1};
  } else if (Wo > 8) {
    c[nc++] = {16, 16, 2, 18, 0};
    c[nc++] = {16, 16, 1, 18, 1};
    c[nc++] = {8, 8, 2, 12, 2};
  } else {
    c[nc++] = {8, 8, 4, 12, 1};   // pitch 12: see lane_pos
    c[nc++] = {8, 8, 2, 12, 2};
  }
  return nc;
}
inline int tiles_of(const Cand& c, int B, int Ho, int Wo, int cout) {
  return Ho > 0 && Wo > 0 ? ((B + c.imgs - 1) / c.imgs) * ((Ho + c.th - 1) / c.th) * ((Wo + c.tw - 1) / c.tw) * (cout / 64) : 0;
}
inline Cand pick_stride1(int B, int Ho, int Wo, int cout, int min_tiles, bool fit = true) {
  Cand c[kMaxCands];
  const int nc = stride1_candidates(Ho, Wo, c, fit);
  for (int i = 0; i < nc; ++i)
    if (tiles_of(c[i], B, Ho, Wo, cout) >= min_tiles) return c[i];
  return c[nc - 1];
}
// stride-2 kernel: 128-pixel tiles (8 x 16, or 8 x 8 x 2 images); window = (2 TH + 1) x (2 TW + 1), columns split by parity
template <class P> inline void set_stride2_geometry(P& p, int Ho, int Wo) {
  if (Wo > 8) { p.TH = 8; p.TW = 16; p.IMGS = 1; }
  else { p.TH = 8; p.TW = 8; p.IMGS = 2; }
  p.HR = 2 * p.TH + 1; p.HC = 2 * p.TW + 1; p.HPH = p.TW + 1; p.HP = 2 * p.HPH;
  p.WTAIL = 0; p.FIT = 0; p.IP = p.HR * p.HP;
  p.tiles_y = (Ho + p.TH - 1) / p.TH; p.tiles_x = (Wo + p.TW - 1) / p.TW;
}
template <class P> inline void set_stride1_geometry(P& p, const Cand& c, int Ho, int Wo) {
  p.TH = c.th; p.TW = c.tw; p.IMGS = c.imgs; p.HP = c.hp; p.HPH = 0;
  p.HR = c.hr ? c.hr : p.TH + 2; p.HC = p.TW + 2; p.WTAIL = c.wtail; p.FIT = c.fit; p.IP = c.ip ? c.ip : p.HR * p.HP;
  p.tiles_y = (Ho + p.TH - 1) / p.TH; p.tiles_x = (Wo + p.TW - 1) / p.TW;
}

// the wide stride-2 variant (conv3x3.inc, HALF): 256-pixel tiles, 8 x 32 output pixels, 16 x 16, or 8 x 8 of four images
template <class P> inline bool set_stride2_wide_geometry(P& p, int Ho, int Wo) {
  p.IMGS = 1;
  if (Wo > 16) { p.TH = 8; p.TW = 32; }
  else if (Wo > 8) { p.TH = 16; p.TW = 16; }
  else if (Wo > 4) { p.TH = 8; p.TW = 8; p.IMGS = 4; }   // four images per tile
  else return false;
  p.WTAIL = 0; p.FIT = 0;
  p.HR = 2 * p.TH + 1; p.HC = 2 * p.TW + 1; p.HPH = p.TW + 1; p.HP = 2 * p.HPH;
  // 8 x 8 maps: the odd-column plane has one column fewer than the even one; without the dead column the four windows are 4 x 17 x 17 pixels =
  // 37 KB and 2 x (40 + 37) KB + tables fit 160 KB (with it: 162 KB).  The price: a row pitch of 34 pixels puts two of a 16-lane group's
  // pixels on one bank group (2-way conflicts on the window fragment reads) -- measured worth it: layer 4's stride-2 conv on the wide kernel
  // is +2.0 % on the whole slide (209.0 k -> 213.2 k patches/s, three passes, one box)
  if (p.TW == 8) p.HP = 2 * p.HPH - 1;
  p.IP = p.HR * p.HP;
  p.tiles_y = (Ho + p.TH - 1) / p.TH; p.tiles_x = (Wo + p.TW - 1) / p.TW;
  return true;
}

// FIT tiles: which pixel of the tile (q = (img * TH + ty) * TW + tx, or -1) each slot of the workgroup's n-tiles computes.  A slot is
// (n-tile t, lane position ll < 32); a ds_read_b128 of a half-wave is served in two groups of 16 lanes, A = {0-3, 12-15, 20-27} and
// B = {4-11, 16-19, 28-31}, and the swizzled window image is read without bank conflicts when the 16 pixels of a group have 16 different
// window indices mod 16 (the nine taps shift all of them alike).  Phase 1 deals every group at most one pixel per residue, richest
// residues first; phase 2 puts what is left (residues that occur more often than there are groups: 7-pixel rows at pitch 8 never produce
// residues 7 and 15) into the free lanes group by group, so that the unavoidable 2-way conflicts sit in as few groups as possible.
inline std::vector<int> deal_fit_pixels(int imgs, int th, int tw, int ip, int hp, int ntile32) {
  static const int ga[16] = {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27};
  static const int gb[16] = {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31};
  std::vector<int> slot((size_t)ntile32 * 32, -1);
  std::vector<std::vector<int>> bucket(16);
  for (int q = imgs * th * tw - 1; q >= 0; --q) {   // (reversed: pop_back hands them out in increasing order)
    const int img = q / (th * tw), ty = (q % (th * tw)) / tw, tx = q % tw;
    bucket[(img * ip + ty * hp + tx) & 15].push_back(q);
  }
  const int ngroups = ntile32 * 2;
  std::vector<std::array<bool, 16>> used((size_t)ngroups);
  std::vector<int> fill((size_t)ngroups, 0);
  auto lane_of = [&](int g, int k) { return (g >> 1) * 32 + ((g & 1) ? gb[k] : ga[k]); };
  for (int g = 0; g < ngroups; ++g) {
    used[g].fill(false);
    int order[16];
    for (int r = 0; r < 16; ++r) order[r] = r;
    std::stable_sort(order, order + 16, [&](int a, int b) { return bucket[a].size() > bucket[b].size(); });
    for (int i = 0; i < 16; ++i) {
      const int r = order[i];
      if (bucket[r].empty()) continue;
      slot[lane_of(g, fill[g]++)] = bucket[r].back();
      bucket[r].pop_back();
      used[g][r] = true;
    }
  }
  for (int g = 0; g < ngroups; ++g)      // phase 2: leftovers, concentrated
    for (int r = 0; r < 16 && fill[g] < 16; ++r)
      if (!bucket[r].empty()) { slot[lane_of(g, fill[g]++)] = bucket[r].back(); bucket[r].pop_back(); }
  for (int g = 0; g < ngroups; ++g)      // (a residue richer than two per group: keep filling)
    for (int r = 0; r < 16; ++r)
      while (!bucket[r].empty() && fill[g] < 16) { slot[lane_of(g, fill[g]++)] = bucket[r].back(); bucket[r].pop_back(); }
  return slot;
}

// Returns nullptr, or the reason the shape cannot be scheduled.  `grid_override` > 0: the launch's share of a merged launch.
// HALF: half-chunk stages of the wide stride-2 variant -- a window piece (1 KiB) is 32 pixels x 32 bytes (lane l: pixel l >> 1, landing
// slot l & 1, which holds global slot (l & 1) ^ ((pixel >> 3) & 1)); waves pair up on pixels although each owns two cout tiles.
template <int STRIDE, int NT, int WAVES, int ESZ, int MT, class P, bool HALF = false>
const char* build_tables(const P& p, int ncb, int grid_override, HostTables* out) {
  constexpr int MAXJ = max_window_pieces<STRIDE, NT, WAVES>();
  const int threads = WAVES * 64, stride = 3 * NT + MAXJ;
  // Invariants the KERNEL relies on and cannot check, stated where its addresses are made:
  //  * at least two channel stages per tile: the mask word of the NEXT tile is fetched by the cursor step that enters the second stage
  //    of the current one (conv3x3.inc, `if (L_ch == 1) fetch_mask()`); with a single stage every tile after the first would reuse tile
  //    0's mask -- wrong padding and out-of-image DMA sources;
  //  * the staged window fits 64 KiB: the hoisted per-lane fragment offsets are kept as 16-bit halves (conv3x3.inc, HOIST), and a DMA
  //    instruction lands 1 KiB, so n_win_instr KiB must cover the window and stay <= 64;
  //  * a descriptor word holds the cout block in bits 0-15, `valid` in bit 16 and the mask row in bits 20-31.
  constexpr int STAGE_PX_BYTES = HALF ? kChunkBytes / 2 : kChunkBytes;
  if ((int64_t)p.Cin * ESZ < 2 * STAGE_PX_BYTES || ((int64_t)p.Cin * ESZ) % STAGE_PX_BYTES) return "conv3x3: needs at least two whole channel stages per tile";
  const int64_t win_px = (int64_t)p.IMGS * p.HR * p.HP + p.WTAIL;
  const int IP = p.IP ? p.IP : p.HR * p.HP;   // pixels between image segments (0 = unset: dense)
  if (win_px * STAGE_PX_BYTES > 65536) return "conv3x3: staged window larger than 64 KiB (16-bit fragment offsets)";
  if ((int64_t)p.n_win_instr * 1024 < win_px * STAGE_PX_BYTES) return "conv3x3: the DMA plan does not cover the staged window";
  constexpr int NTILE32 = (HALF ? WAVES / 2 : WAVES * MT / 2) * NT;   // 32-pixel n-tiles of a workgroup
  if (p.FIT) {
    if (STRIDE != 1 || HALF) return "conv3x3: fit tiles are stride-1 tiles";
    if (p.IMGS * p.TH * p.TW > NTILE32 * 32) return "conv3x3: fit tile larger than the workgroup's slots";
    // shared halos are zero only where they lie outside the image: a shared column needs full-width tiles, a shared row whole images
    if (p.HP < p.TW + 1 || p.HR < p.TH + 1) return "conv3x3: fit tile pitch too small";
    if (p.HP == p.TW + 1 && (p.TW != p.Wo || p.tiles_x != 1)) return "conv3x3: a shared halo column needs tiles of the map's full width";
    if (p.HR == p.TH + 1 && (p.TH != p.Ho || p.tiles_y != 1)) return "conv3x3: a shared halo row needs tiles of the map's full height";
    if (IP < p.HR * p.HP) return "conv3x3: fit tile segments overlap";
    // the last tap of the last pixel reads window index (IMGS - 1) IP + (TH + 1) HP + TW + 1
    if ((int64_t)(p.IMGS - 1) * IP + (int64_t)(p.TH + 1) * p.HP + p.TW + 1 >= win_px) return "conv3x3: fit tile window tail too short";
  } else if (p.IMGS * p.TH * p.TW != NTILE32 * 32 || p.WTAIL != 0 || p.HR != STRIDE * p.TH + (STRIDE == 2 ? 1 : 2) || IP != p.HR * p.HP) {
    return "conv3x3: tile / pixel mismatch";
  }
  const std::vector<int> fit_slot = p.FIT ? deal_fit_pixels(p.IMGS, p.TH, p.TW, IP, p.HP, NTILE32) : std::vector<int>();
  if (ncb < 1 || ncb > 65535) return "conv3x3: cout block count does not fit the descriptor word";
  std::vector<int>& lane = out->lane;
  lane.assign((size_t)threads * stride, 0);
  // per-thread geometry kept for the mask rows below: output pixel (ty, tx, img) per n-tile, window piece (hy, hx, img, live)
  std::vector<std::array<int, 3>> pix((size_t)threads * NT);
  std::vector<std::array<int, 4>> win((size_t)threads * MAXJ);
  const int64_t img_in_bytes = (int64_t)p.Hi * p.Wi * p.Cin * ESZ;
  for (int tid = 0; tid < threads; ++tid) {
    const int l = tid & 63, wave = tid >> 6;
    int* row = &lane[(size_t)tid * stride];
    // Lane -> pixel of the n-tile.  A ds_read_b128 is served in two groups of 16 lanes per half-wave,
    // A = {0-3, 12-15, 20-27} and B = {4-11, 16-19, 28-31}; the swizzled window image is conflict-free when the 16
    // pixels of a group have 16 different (linear index mod 16).  Rows of 32 pixels satisfy that in lane order;
    // 16- and 8-pixel rows do when group A takes rows {0} / {0, 2} and group B rows {1} / {1, 3} (row pitches are
    // chosen so that those row pairs cover disjoint residues).
    auto lane_pos = [&](int ll) {
      if (p.TW >= 32) return ll;
      const bool in_a = ll < 4 || (ll >= 12 && ll < 16) || (ll >= 20 && ll < 28);
      const int rank = in_a ? (ll < 4 ? ll : ll < 16 ? ll - 8 : ll - 12) : (ll < 12 ? ll - 4 : ll < 20 ? ll - 8 : ll - 16);
      if (p.TW == 16) return (in_a ? 0 : 16) + rank;
      return (in_a ? 0 : 8) + (rank < 8 ? rank : rank + 8);   // TW == 8: A -> rows 0, 2; B -> rows 1, 3
    };
    for (int nt = 0; nt < NT; ++nt) {
      const int tile32 = (MT == 2 && !HALF ? wave : wave >> 1) * NT + nt;   // MT == 1 / HALF: wave pairs share pixels
      int pidx = tile32 * 32 + lane_pos(l & 31);
      if (p.FIT) pidx = fit_slot[(size_t)tile32 * 32 + (l & 31)];
      if (pidx < 0) {   // FIT: a slot without a pixel -- never valid (image index IMGS), addresses harmless
        pix[(size_t)tid * NT + nt] = {0, 0, p.IMGS};
        row[nt] = 0; row[NT + nt] = 0; row[2 * NT + nt] = 0;
        continue;
      }
      const int img = pidx / (p.TH * p.TW), rem = pidx % (p.TH * p.TW);
      const int ty = rem / p.TW, tx = rem % p.TW;
      pix[(size_t)tid * NT + nt] = {ty, tx, img};
      const int64_t orel = (int64_t)img * p.o_img + (int64_t)ty * p.o_row + (int64_t)tx * p.o_px;   // relative to the tile's origin
      if (orel < 0 || orel > INT32_MAX) return "conv3x3: per-lane output offset does not fit 31 bits";
      row[nt] = (int)orel;
      row[NT + nt] = img * IP + ty * STRIDE * p.HP + tx;
      const int64_t rrel = (int64_t)img * p.o_img + (int64_t)ty * p.r_row + (int64_t)tx * p.r_px;   // the residual's own layout
      if (rrel < 0 || rrel > INT32_MAX) return "conv3x3: per-lane residual offset does not fit 31 bits";
      row[2 * NT + nt] = (int)rrel;
    }
    // window DMA: instruction i = wave + WAVES*j fills LDS pixels 16i..16i+15; lane l fills LDS slot (l&3)
    // of pixel 16i + l/4 with GLOBAL slot (l&3) ^ swizzle(pixel)
    for (int j = 0; j < MAXJ; ++j) {
      const int i = wave + WAVES * j;
      const int px = HALF ? i * 32 + (l >> 1) : i * 16 + (l >> 2);
      const int img = px / IP, r = px % IP;   // (r >= HR * HP: a gap pixel between the segments of a fit window -- never staged)
      const int hy = r / p.HP, c = r % p.HP;
      int hx = c;
      if (STRIDE == 2) hx = 2 * (c % p.HPH) + c / p.HPH;
      const bool live = i < p.n_win_instr && img < p.IMGS && hy < p.HR && hx < p.HC;
      const int g = HALF ? (l & 1) ^ ((px >> 3) & 1) : (l & 3) ^ ((px >> 2) & 3);
      const int64_t roff = (int64_t)img * img_in_bytes + ((int64_t)(hy - 1) * p.Wi + hx - 1) * p.in_px_bytes + g * 16;
      if (live && (roff < INT32_MIN || roff > INT32_MAX)) return "conv3x3: per-lane window offset does not fit 32 bits";
      row[3 * NT + j] = live ? (int)roff : 0;   // (dead pieces are never dereferenced: their mask bit is clear in every row)
      win[(size_t)tid * MAXJ + j] = {hy, hx, img, live ? 1 : 0};
    }
  }
  // Mask rows: one per distinct (oy0, ox0, images left in the group): bit j = window piece j of the thread lies inside the
  // image, bit 16 + nt = output pixel nt exists.  The kernel fetches its word of the row one tile ahead.
  std::map<std::array<int, 3>, int> mask_row;
  std::vector<unsigned>& mask = out->mask;
  mask.clear();
  auto mask_row_of = [&](int img0, int oy0, int ox0) {
    const std::array<int, 3> mk = {oy0, ox0, std::min(p.IMGS, std::max(0, p.B - img0))};
    auto it = mask_row.find(mk);
    if (it != mask_row.end()) return it->second;
    const int r = (int)mask_row.size();
    mask_row[mk] = r;
    mask.resize((size_t)(r + 1) * threads);
    for (int tid = 0; tid < threads; ++tid) {
      unsigned m = 0;
      for (int j = 0; j < MAXJ; ++j) {
        const auto& w = win[(size_t)tid * MAXJ + j];
        const int iy = oy0 * STRIDE + w[0] - 1, ix = ox0 * STRIDE + w[1] - 1;
        if (w[3] && w[2] < mk[2] && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi) m |= 1u << j;
      }
      for (int nt = 0; nt < NT; ++nt) {
        const auto& q = pix[(size_t)tid * NT + nt];
        if (q[2] < mk[2] && oy0 + q[0] < p.Ho && ox0 + q[1] < p.Wo) m |= 1u << (16 + nt);
      }
      mask[(size_t)r * threads + tid] = m;
    }
    return r;
  };
  // Schedule [iters][grid]: which (pixel tile, cout block) a workgroup takes in which iteration.  Workgroups are
  // dispatched round-robin over the 8 XCDs (workgroup w -> XCD w % 8), each with its own 4 MiB L2.  The ncb cout
  // blocks of a pixel tile go to ncb workgroups of ONE XCD in the same iteration (`xcd_group`), so the staged window
  // is fetched into that L2 once and hit ncb - 1 times; measured -8 % (layer 2) ... -16 % (stride-2 layers) against
  // cout block = workgroup % ncb (one weight slice per XCD), also for the 512-channel layers whose 4.7 MB of weights
  // no longer fit one L2 (they come from the Infinity Cache instead).
  // Either way a workgroup keeps its cout block for all iterations (resident-weight variants rely on it).
  const int tiles_per_img = p.tiles_y * p.tiles_x;
  const int grid = grid_override > 0 ? grid_override : std::min(256, p.ntiles), n_pt = p.ntiles / ncb;   // (override: a share of a merged launch)
  const bool xcd_group = ncb > 1 && grid == 256 && 32 % ncb == 0 && (int64_t)p.Cout * p.Cin * 9 * ESZ <= (1 << 23);
  std::vector<TileDesc>& tile = out->tile;
  tile.assign((size_t)p.iters * grid, TileDesc{0, 0, 0, 0});
  for (int it = 0; it < p.iters; ++it)
    for (int w = 0; w < grid; ++w) {
      int cb, pt;
      if (xcd_group) {
        const int xcd = w % 8, slot = w / 8;
        cb = slot % ncb;
        pt = it * (grid / ncb) + (slot / ncb) * 8 + xcd;
      } else {
        const int T_ = it * grid + w;
        cb = T_ % ncb; pt = T_ / ncb;
      }
      const int valid = pt < n_pt ? 1 : 0, t = pt % tiles_per_img;
      const int img0 = valid ? (pt / tiles_per_img) * p.IMGS : 0, oy0 = (t / p.tiles_x) * p.TH, ox0 = (t % p.tiles_x) * p.TW;
      const int mrow = mask_row_of(img0, oy0, ox0);
      if (mrow >= 4096) return "conv3x3: too many distinct tile positions for the mask table";
      const int64_t win_off = (int64_t)img0 * img_in_bytes + ((int64_t)(oy0 * STRIDE) * p.Wi + ox0 * STRIDE) * p.in_px_bytes;
      const int64_t out_off = (int64_t)img0 * p.o_img + (int64_t)oy0 * p.o_row + (int64_t)ox0 * p.o_px + p.o_base + (int64_t)cb * p.out_cb;
      const int64_t res_off = (int64_t)img0 * p.o_img + (int64_t)oy0 * p.r_row + (int64_t)ox0 * p.r_px + p.r_base + (int64_t)cb * p.r_cb;
      if (win_off >= ((int64_t)1 << 32) || out_off >= ((int64_t)1 << 32) || res_off >= ((int64_t)1 << 32)) return "conv3x3: tensor larger than 4 Gi elements / bytes";
      const uint32_t word = (uint32_t)cb | (uint32_t)valid << 16 | (uint32_t)mrow << 20;   // unsigned: mask rows >= 2048 reach bit 31
      tile[(size_t)it * grid + w] = TileDesc{(int)word, (int)(uint32_t)win_off, (int)(uint32_t)out_off, (int)(uint32_t)res_off};
    }
  out->grid = grid; out->threads = threads; out->lane_stride = stride; out->mask_rows = (int)mask_row.size(); out->xcd_group = xcd_group;
  return nullptr;
}

}  // namespace dh_conv3
