// opk_attn.hip.h -- attention on fragment-packed q / k / v^T
#pragma once

#include <type_traits>

#include "opk_common.hip.h"

namespace opk {

// ----------------------------------------------------------------------------------------------
// Attention on fragment-packed q / k / v^T (row-stationary path).  Same algorithm as attn_kernel above
// (transposed scores, lane-local online softmax, O^T = V^T P^T) but every byte moves as 1 KiB pieces:
//   * Q fragments: one coalesced 16-byte-per-lane load per (k-step, plane), straight into registers;
//   * K / V^T tiles: 32 pieces per 64-key tile (16 KiB hi + 16 KiB lo) copied global -> LDS by DMA
//     (global_load_lds), double-buffered, fragment reads lane-linear and conflict-free;
//   * output: the v^T piece order was chosen at QKV time so that a lane ends up with 8 consecutive head
//     dims -> two 16-byte stores per plane per lane, whole 1 KiB pieces of the fragment-packed o.
// Sequences start at multiples of 32 rows, so key tiles coincide with whole pieces.
// ----------------------------------------------------------------------------------------------
struct AttnFpParams {
  const u16* q_fp;   // [rows/16][H/32][2][512]
  const u16* k_fp;
  const u16* vt_fp;  // [heads][rows/32][2][4][512]
  u16* o_fp;         // [rows/16][H/32][2][512]; OF8: fp16 pieces [rows/16][H/32][512] ...
  u16* o_lo8;        // ... and e4m3 lo pieces [rows/16][heads][512] (one 1 KiB half-fragment per head, opk_common.hip.h)
  const int32_t* cu;
  int s0;
  const int32_t* roff;
  const int32_t* qboff;  // first work item (query block) of each sequence, [ns + 1]: the grid has no empty blocks
  int ns;
  int H;
  int r_pad;
  int window;
  int n_items;  // work items (sequence, query block) per head
  int n_heads;
  int xcd_group;  // 0: item-major grid (n_items x n_heads blocks); 1: XCD-grouped map below
  int* range_flag;  // OF8 + H16 (kernel sets 10 / 11): raised when an output is not finite (see PanelParams::range_flag); may be NULL
};

// Block -> (work item, head), XCD-aware.  Neighbouring query blocks of a sequence read the same K / V^T rows (a
// sliding-window block of 128 queries reads 256 keys, a full-attention block all of them), so they should share an L2:
// block b is dispatched to XCD b % 8 (observed; a speed assumption only), and with a one-dimensional grid of
// ceil(n_items / 32) * 32 * n_heads blocks
//   group of 4 consecutive items = (b / 8 / (4 * n_heads)) * 8 + b % 8,  head = (b / 8 / 4) % n_heads,  item = 4 * group + (b / 8) % 4
// the four query blocks of a 512-token sequence (same head) follow each other on one XCD.  Row-major (item, head) grids
// spread them over four XCDs and fetched K / V^T twice.  Measured (same box, A/B): base model (H = 512, 8 heads, panel
// path) sliding-window attention -12 %, full attention fetches -40 %; xsmall (H = 256, row path) +1 % on attention and
// +1 % on the whole-layer kernel that follows -- its 128-row block i runs on XCD i % 8, the XCD the item-major map lets
// write those rows of o, and its q / k / v planes mostly sit in the Infinity Cache anyway.  The host therefore selects
// the grouped map on the panel path only (xcd_group).
constexpr int ATT_ITEM_GROUP = 4;

// queries per block of the fragment-packed attention kernel = WAVES x 32 (4 waves: two or more blocks per CU; 8 waves:
// one block per CU, every K / V^T tile staged once for 256 queries -> half the DMA instructions and L2 traffic).
// Keys per tile = 32 x KT.  Full-attention layers use KT = 2 (64-key tiles); sliding-window layers KT = 1: a wave of
// 32 queries at qbase sees the keys [qbase - w, qbase + 31 + w] -- with w = 64 exactly five aligned 32-key tiles, of
// which the three inner ones are mask-free and only the two outer ones carry the window's diagonal -- where 64-key
// tiles made it walk 192..256 keys, nearly all of them through the masked path.
//
// Term masks: TQK for S^T = K Q^T (left = q, right = k), TPV for O^T = V^T P^T (left = p, right = v); O_LO: the
// output o also gets a lo plane (it is the left operand of the attention output projection).
//
// Scores arrive pre-multiplied by log2(e) (folded into the q scale by the QKV epilogue), so the softmax uses
// exp2 directly: p = 2^(s - max).
// ZP: lo(p) is cleared (a policy without the lo(p) x hi(v) term evaluated on the instantiation that has it).
// OF8: the output is written for the "f16 + fp8" whole-layer kernel: hi = fp16 pieces, lo = e4m3 x 2^12 (a head's 64
// dims are one 16-byte half-fragment per lane: the lane's 8 dims of both k-steps of the head).
// NST: LDS stages of the K / V^T tile ring.  2 (shipped): the next tile is requested while this one is multiplied.
// 3 (experiment, see op_launch_attn.hip: slower): two tiles ahead, the end-of-tile wait counts the newest request out
// (vmcnt retires in order) instead of draining everything.
// H16 (kernel set "f16"): q / k / v^T / p / o are single-plane fp16 (the layouts of the single-pass bf16 set), products on
// v_mfma_f32_16x16x32_f16.  p <= 2^6 by the lazy reference; a p below 2^-24 is flushed (it is summed into l_run in fp32).
template <int TQK, int TPV, bool O_LO, int WAVES, int KT, bool ZP = false, bool OF8 = false, int NST = 2, bool H16 = false>
__global__ __launch_bounds__(WAVES * 64, 2) void attn_fp_kernel(AttnFpParams p) {
  if constexpr (OF8 && !H16) set_saturating_conversions();  // (H16: only in front of the output conversions, below)
  static_assert(!H16 || (TQK == 0 && TPV == 0 && !O_LO && !ZP), "fp16 operands: single pass only");
  constexpr int ATT_FP_BQ = WAVES * 32;
  constexpr int TILE_KEYS = 32 * KT;
  constexpr bool Q_LO = (TQK & T_LEFT_LO) != 0, K_LO = (TQK & T_RIGHT_LO) != 0;
  constexpr bool P_LO = (TPV & T_LEFT_LO) != 0, V_LO = (TPV & T_RIGHT_LO) != 0;
  constexpr int PK = K_LO ? 2 : 1, PV = V_LO ? 2 : 1;
  constexpr int K_PIECES = 4 * KT * PK;             // [m 0..2KT-1][ks 0..1][plane]
  constexpr int V_PIECES = 4 * KT * PV;             // [t 0..KT-1][plane][n 0..3]
  constexpr int STAGE = (K_PIECES + V_PIECES) * 512;
  static_assert(K_PIECES % WAVES == 0 && V_PIECES % WAVES == 0, "tile pieces must split evenly over the waves");
  static_assert(NST == 2 || NST == 3, "two or three tile stages");
  __shared__ __attribute__((aligned(16))) u16 sT[NST][STAGE];

  // work item -> (sequence, query block): binary search in the per-sequence prefix of ceil(len / ATT_FP_BQ)
  const int xcd_slot = blockIdx.x >> 3;
  const int item = p.xcd_group
                       ? ((xcd_slot / (ATT_ITEM_GROUP * p.n_heads)) * 8 + (blockIdx.x & 7)) * ATT_ITEM_GROUP + xcd_slot % ATT_ITEM_GROUP
                       : (int)(blockIdx.x % (unsigned)p.n_items);
  if (item >= p.n_items) return;  // the grid is rounded up to whole groups on every XCD
  int s = 0;
  {
    int lo_s = 0, hi_s = p.ns - 1;
    while (lo_s < hi_s) {
      const int mid = (lo_s + hi_s + 1) >> 1;
      if (p.qboff[mid] <= item) lo_s = mid; else hi_s = mid - 1;
    }
    s = lo_s;
  }
  const int head = p.xcd_group ? (xcd_slot / ATT_ITEM_GROUP) % p.n_heads : (int)(blockIdx.x / (unsigned)p.n_items);
  const int q0 = (item - p.qboff[s]) * ATT_FP_BQ;
  const int len = p.cu[p.s0 + s + 1] - p.cu[p.s0 + s];
  if (q0 >= len) return;
  const int r0 = p.roff[s];
  const int alloc = p.roff[s + 1] - r0;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15;
  const int g = lane >> 4;
  const int kbn = p.H >> 5;  // k-steps per row block

  const int qbase = q0 + wave * 32;   // this wave's 32 queries = two fragments
  const bool active = qbase < alloc;  // alloc is a multiple of 32: both fragments in or out
  const size_t q_rb = (size_t)((r0 + (active ? qbase : q0)) >> 4);

  bf16x8 qf_hi[2][2], qf_lo[2][2];  // [query fragment][k-step]
#pragma unroll
  for (int qf = 0; qf < 2; ++qf)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const u16* src = p.q_fp + (((q_rb + qf) * kbn + head * 2 + ks) * 2) * 512 + lane * 8;
      qf_hi[qf][ks] = load_stream_frag(src);
      qf_lo[qf][ks] = Q_LO ? load_stream_frag(src + 512) : qf_hi[qf][ks];
    }
#pragma unroll
  for (int qf = 0; qf < 2; ++qf)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {  // retire the loads before the tile loop (see rowgemm_kernel)
      asm volatile("" : "+v"(qf_hi[qf][ks]));
      asm volatile("" : "+v"(qf_lo[qf][ks]));
    }

  const int win = p.window >= 0 ? p.window : (1 << 30);
  int kt_lo = 0, kt_hi = (len - 1) / TILE_KEYS;
  if (p.window >= 0) {
    const int lo_key = q0 - p.window;
    kt_lo = lo_key > 0 ? lo_key / TILE_KEYS : 0;
    const int hi_t = (q0 + ATT_FP_BQ - 1 + p.window) / TILE_KEYS;
    kt_hi = hi_t < kt_hi ? hi_t : kt_hi;
  }

  // DMA one key tile: wave w copies pieces w, w + WAVES, ... of the stage [K: (m, ks, plane) | V^T: (t, plane, n)].
  // Everything about a piece except the tile's row block is wave-constant and computed once, in 32-bit element
  // offsets (the buffers are < 2^31 elements): per tile a piece costs one clamp, one multiply and one add.
  const int k_rb_max = (p.r_pad >> 4) - 1, v_tb_max = (p.r_pad >> 5) - 1;
  const int k_stride = kbn * 1024;  // elements per 16-row block of k
  constexpr int KPW = K_PIECES / WAVES, VPW = V_PIECES / WAVES;
  int k_m[KPW], k_off[KPW], v_t[VPW], v_off[VPW];
#pragma unroll
  for (int u = 0; u < KPW; ++u) {
    const int kp = wave + WAVES * u;
    const int rem = kp % (2 * PK);
    k_m[u] = kp / (2 * PK);
    k_off[u] = ((head * 2 + rem / PK) * 2 + rem % PK) * 512;
  }
#pragma unroll
  for (int u = 0; u < VPW; ++u) {
    const int vp = wave + WAVES * u;
    const int rem = vp % (4 * PV);
    v_t[u] = vp / (4 * PV);
    v_off[u] = head * (p.r_pad >> 5) * 4096 + (rem / 4) * 2048 + (rem % 4) * 512;
  }
  auto stage_tile = [&](int kt, int stage) {
    // A tile may reach past the last computed row (the last sequence need not fill its final tile): such
    // pieces are clamped onto the last valid one -- their keys are masked, they only have to be finite.
    const int k_rb0 = (r0 + kt * TILE_KEYS) >> 4;
    const int v_tb0 = (r0 + kt * TILE_KEYS) >> 5;
#pragma unroll
    for (int u = 0; u < KPW; ++u) {
      int rb = k_rb0 + k_m[u];
      rb = rb < k_rb_max ? rb : k_rb_max;
      const u16* src = p.k_fp + (unsigned)(rb * k_stride + k_off[u]);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 8),
                                       (__attribute__((address_space(3))) void*)(&sT[stage][(wave + WAVES * u) * 512]), 16,
                                       0, 0);
    }
#pragma unroll
    for (int u = 0; u < VPW; ++u) {
      int tb = v_tb0 + v_t[u];
      tb = tb < v_tb_max ? tb : v_tb_max;
      const u16* src = p.vt_fp + (unsigned)(tb * 4096 + v_off[u]);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + lane * 8),
          (__attribute__((address_space(3))) void*)(&sT[stage][(K_PIECES + wave + WAVES * u) * 512]), 16, 0, 0);
    }
  };

  // FSM (round 5; the single-pass instantiations, where the softmax's vector instructions -- not the MFMAs -- bound the
  // kernel: ~150 of them per 64-key tile and wave beside 32 MFMAs): two of the five vector instructions per score leave.
  //   * The score accumulators start at -m_run instead of 0, so S^T arrives as s - m and goes straight into v_exp (no
  //     subtraction); the reference still moves lazily, and only a tile in which it moves pays the subtraction.
  //   * The row sums come from the matrix pipe: one more MFMA per k-step against a fragment of ones accumulates
  //     sum_k p[k][q] -- of the ROUNDED p the numerator is made of -- for every lane of a query column (no adds, no shuffles).
  constexpr bool FSM = H16;  // (not the bf16 single-pass set: it stays bit-identical to the all-terms kernels with cleared lo operands)
  float m_run[2] = {-1e30f, -1e30f};
  float l_run[2] = {0.f, 0.f};
  f32x4 lacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};  // FSM: row sums (all four rows of a lane hold the same sum)
  const bf16x8 ones_frag = as_frag(H16 ? make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u)
                                       : make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u));
  f32x4 oacc[4][2];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int qf = 0; qf < 2; ++qf) oacc[n][qf] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto tile = [&](int kt, auto cur_tag) {
    constexpr int cur = decltype(cur_tag)::value;
    // unconditional prefetch into the idle stage (NST = 3: of the tile after next, into the stage tile kt - 1 used)
    if constexpr (NST == 2) stage_tile(kt + 1 <= kt_hi ? kt + 1 : kt, cur ^ 1);
    else stage_tile(kt + 2 <= kt_hi ? kt + 2 : kt_hi, (cur + 2) % 3);
    const u16* st = &sT[cur][lane * 8];
    const int kbase = kt * TILE_KEYS;
    // wave-uniform tile classification: skip tiles entirely outside this wave's window (other waves of the block
    // may need them), and use the mask-free path when every (query, key) pair of the tile is visible
    const bool outside = (kbase > qbase + 31 + win) || (kbase + TILE_KEYS - 1 < qbase - win);
    const bool all_valid =
        (kbase + TILE_KEYS - 1 < len) && (kbase + TILE_KEYS - 1 - qbase <= win) && (qbase + 31 - kbase <= win);
    if (!outside) {
      f32x4 sacc[2 * KT][2];
      // FSM: the accumulators start at minus the softmax reference of their query (0 while it has none: sentinel -1e30)
      float m_base[2];
#pragma unroll
      for (int qf = 0; qf < 2; ++qf) m_base[qf] = (FSM && m_run[qf] > -1e29f) ? m_run[qf] : 0.f;
#pragma unroll
      for (int m = 0; m < 2 * KT; ++m)
#pragma unroll
        for (int qf = 0; qf < 2; ++qf) sacc[m][qf] = f32x4{-m_base[qf], -m_base[qf], -m_base[qf], -m_base[qf]};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int m = 0; m < 2 * KT; ++m) {
          const bf16x8 kh = lds_frag(st + ((m * 2 + ks) * PK) * 512);
          const bf16x8 kl = K_LO ? lds_frag(st + ((m * 2 + ks) * PK + 1) * 512) : kh;
#pragma unroll
          for (int qf = 0; qf < 2; ++qf) {
            if (K_LO) sacc[m][qf] = mfma16(kl, qf_hi[qf][ks], sacc[m][qf]);
            if (Q_LO) sacc[m][qf] = mfma16(kh, qf_lo[qf][ks], sacc[m][qf]);
            sacc[m][qf] = mfma16x<H16>(kh, qf_hi[qf][ks], sacc[m][qf]);
          }
        }
      }

      if (!all_valid) {
        // masked scores become -3e30 (below the running-max initial value -1e30): 2^(masked - max) underflows to
        // exactly 0 even when the whole tile is masked for a query.  A query sees the keys [lo, hi] =
        // [max(q - win, 0), min(q + win, len - 1)]; element (m, r) of this lane is key kbase + 4g + (16m + r), so
        // with lo_rel = lo - kbase - 4g the test is one unsigned compare of the constant (16m + r) - lo_rel
        // against hi - lo (an empty interval is moved out of reach).
#pragma unroll
        for (int qf = 0; qf < 2; ++qf) {
          const int qpos = qbase + 16 * qf + l15;
          int lo = qpos - win, hi = qpos + win;
          lo = lo > 0 ? lo : 0;
          hi = hi < len - 1 ? hi : len - 1;
          const int span = hi - lo;
          const int lo_rel = span >= 0 ? lo - kbase - 4 * g : (1 << 29);
          const unsigned uspan = span >= 0 ? (unsigned)span : 0u;
#pragma unroll
          for (int m = 0; m < 2 * KT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool ok = (unsigned)(16 * m + r - lo_rel) <= uspan;
              sacc[m][qf][r] = ok ? sacc[m][qf][r] : -3e30f;
            }
        }
      }

      bf16x8 ph[KT][2], pl[KT][2];  // [k-step t][query fragment]
#pragma unroll
      for (int qf = 0; qf < 2; ++qf) {
        float tile_max = -3e30f;
#pragma unroll
        for (int m = 0; m < 2 * KT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) tile_max = fmaxf(tile_max, sacc[m][qf][r]);
        tile_max = fmaxf(tile_max, __shfl_xor(tile_max, 16, 64));
        tile_max = fmaxf(tile_max, __shfl_xor(tile_max, 32, 64));
        if constexpr (FSM) {
          // sacc holds s - m_base.  The reference moves when the tile's maximum exceeds it by more than 2^6 (as below), or
          // when this query has no reference yet and the tile has a visible key for it (a masked tile leaves -3e30).
          const bool fresh = m_run[qf] <= -1e29f;
          const bool moved = fresh ? (tile_max > -1e29f) : (tile_max > 6.0f);
          if (__builtin_amdgcn_ballot_w64(moved) != 0) {  // wave-uniform, rare after a query's first tile
            const float delta = moved ? tile_max : 0.f;
            const float alpha = __builtin_amdgcn_exp2f(fresh ? 0.f : -delta);  // (a fresh query's sums are zero: any finite factor)
            m_run[qf] = moved ? m_base[qf] + delta : m_run[qf];
            lacc[qf][0] *= alpha;
            lacc[qf][1] *= alpha;
            lacc[qf][2] *= alpha;
            lacc[qf][3] *= alpha;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
              oacc[n][qf][0] *= alpha;
              oacc[n][qf][1] *= alpha;
              oacc[n][qf][2] *= alpha;
              oacc[n][qf][3] *= alpha;
            }
#pragma unroll
            for (int m = 0; m < 2 * KT; ++m)
#pragma unroll
              for (int r = 0; r < 4; ++r) sacc[m][qf][r] -= delta;
          }
#pragma unroll
          for (int m = 0; m < 2 * KT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[m][qf][r] = __builtin_amdgcn_exp2f(sacc[m][qf][r]);  // <= 2^6; masked: 2^-3e30 = 0
#pragma unroll
          for (int t = 0; t < KT; ++t) {
            const float v0[4] = {sacc[2 * t][qf][0], sacc[2 * t][qf][1], sacc[2 * t][qf][2], sacc[2 * t][qf][3]};
            const float v1[4] = {sacc[2 * t + 1][qf][0], sacc[2 * t + 1][qf][1], sacc[2 * t + 1][qf][2], sacc[2 * t + 1][qf][3]};
            uint2 h0, l0, h1, l1;
            split4x<false, H16>(v0, h0, l0);
            split4x<false, H16>(v1, h1, l1);
            ph[t][qf] = as_frag(make_uint4(h0.x, h0.y, h1.x, h1.y));
            pl[t][qf] = ph[t][qf];
          }
          continue;
        }
        // Lazy reference: the exponent reference m_run only moves when this tile's maximum exceeds it by more
        // than 2^6 (p stays <= 64, harmless in fp32 and in the hi/lo split), and the 16 accumulator rescales run
        // only in the tiles where some query of the wave moved -- a wave-uniform branch, rare after the first tile.
        const bool moved = tile_max > m_run[qf] + 6.0f;
        if (__builtin_amdgcn_ballot_w64(moved) != 0) {
          const float m_new = moved ? tile_max : m_run[qf];
          const float alpha = __builtin_amdgcn_exp2f(m_run[qf] - m_new);
          m_run[qf] = m_new;
          l_run[qf] *= alpha;
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            oacc[n][qf][0] *= alpha;
            oacc[n][qf][1] *= alpha;
            oacc[n][qf][2] *= alpha;
            oacc[n][qf][3] *= alpha;
          }
        }
        const float m_ref = m_run[qf];
        float psum = 0.f;
#pragma unroll
        for (int m = 0; m < 2 * KT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // raw v_exp_f32: the argument is <= 6 and a result below 2^-126 may flush to zero
            const float e = __builtin_amdgcn_exp2f(sacc[m][qf][r] - m_ref);
            sacc[m][qf][r] = e;
            psum += e;
          }
        l_run[qf] += psum;
        // lane slot e < 4 -> key 32t + 4g + e (piece m = 2t), e >= 4 -> key 32t + 16 + 4g + (e-4) (piece 2t+1):
        // the order the QKV epilogue stored v^T in
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          const float v0[4] = {sacc[2 * t][qf][0], sacc[2 * t][qf][1], sacc[2 * t][qf][2], sacc[2 * t][qf][3]};
          const float v1[4] = {sacc[2 * t + 1][qf][0], sacc[2 * t + 1][qf][1], sacc[2 * t + 1][qf][2], sacc[2 * t + 1][qf][3]};
          uint2 h0, l0, h1, l1;
          split4x<P_LO, H16>(v0, h0, l0);
          split4x<P_LO, H16>(v1, h1, l1);
          ph[t][qf] = as_frag(make_uint4(h0.x, h0.y, h1.x, h1.y));
          pl[t][qf] = ZP ? as_frag(make_uint4(0u, 0u, 0u, 0u)) : as_frag(make_uint4(l0.x, l0.y, l1.x, l1.y));
        }
      }

      // O^T += V^T P^T
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        if constexpr (FSM) {  // the row sums of this k-step: ones x P^T
#pragma unroll
          for (int qf = 0; qf < 2; ++qf) lacc[qf] = mfma16x<H16>(ones_frag, ph[t][qf], lacc[qf]);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const bf16x8 vh = lds_frag(st + (K_PIECES + (t * PV) * 4 + n) * 512);
          const bf16x8 vl = V_LO ? lds_frag(st + (K_PIECES + (t * PV + 1) * 4 + n) * 512) : vh;
#pragma unroll
          for (int qf = 0; qf < 2; ++qf) {
            if (V_LO) oacc[n][qf] = mfma16(vl, ph[t][qf], oacc[n][qf]);
            if (P_LO) oacc[n][qf] = mfma16(vh, pl[t][qf], oacc[n][qf]);
            oacc[n][qf] = mfma16x<H16>(vh, ph[t][qf], oacc[n][qf]);
          }
        }
      }
    }
    if constexpr (NST == 2) {
      __syncthreads();
    } else {  // tile kt + 1 has landed (the request for kt + 2, issued at the top, may still fly); then all waves meet
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KPW + VPW) : "memory");
      __builtin_amdgcn_s_barrier();
    }
  };

  stage_tile(kt_lo, 0);
  if constexpr (NST == 2) {
    __syncthreads();
    for (int kt = kt_lo; kt <= kt_hi; kt += 2) {
      tile(kt, std::integral_constant<int, 0>{});
      if (kt + 1 <= kt_hi) tile(kt + 1, std::integral_constant<int, 1>{});
    }
  } else {
    stage_tile(kt_lo + 1 <= kt_hi ? kt_lo + 1 : kt_hi, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KPW + VPW) : "memory");
    __builtin_amdgcn_s_barrier();
    for (int kt = kt_lo; kt <= kt_hi; kt += 3) {
      tile(kt, std::integral_constant<int, 0>{});
      if (kt + 1 <= kt_hi) tile(kt + 1, std::integral_constant<int, 1>{});
      if (kt + 2 <= kt_hi) tile(kt + 2, std::integral_constant<int, 2>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the clamped prefetches of the last tiles)
  }

  if constexpr (OF8 && H16) set_saturating_conversions();  // the e4m3 lo plane of o clamps; p (<= 2^6) was converted with IEEE overflow
  if (active) {
    bool o_bad = false;  // OF8 + H16: an output beyond fp16's range or not finite
#pragma unroll
    for (int qf = 0; qf < 2; ++qf) {
      float l_tot = l_run[qf] + __shfl_xor(l_run[qf], 16, 64);
      l_tot += __shfl_xor(l_tot, 32, 64);
      if constexpr (FSM) l_tot = lacc[qf][0];  // summed over all 32 keys of every k-step by the MFMA: complete in every lane
      const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
      // oacc[n][qf][r]: d = 32(n>>1) + 8g + 4(n&1) + r  ->  lane owns d = 8g..8g+7 of k-step (n>>1) of this head
      if constexpr (OF8) {
        uint32_t lo8[4];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const float v0[4] = {oacc[2 * half][qf][0] * inv, oacc[2 * half][qf][1] * inv, oacc[2 * half][qf][2] * inv,
                               oacc[2 * half][qf][3] * inv};
          const float v1[4] = {oacc[2 * half + 1][qf][0] * inv, oacc[2 * half + 1][qf][1] * inv,
                               oacc[2 * half + 1][qf][2] * inv, oacc[2 * half + 1][qf][3] * inv};
          if constexpr (H16) {
            // Kernel sets 10 / 11: q / k / v^T are fp16 and hold Inf where an activation went beyond fp16's range; the scores and
            // o are NaN then -- but the consumer of o multiplies under MODE.FP16_OVFL = 1, where v_mfma_f32_16x16x32_f16 takes a
            // NaN operand for a finite number and a +Inf - Inf inside the dot product likewise (microbench/mode_probe.hip): the
            // forward would be computed from a swallowed operand.  Raise the range flag instead (rank_head_kernel -> NaN).
#pragma unroll
            for (int r = 0; r < 4; ++r) o_bad |= !(fabsf(v0[r]) <= 65504.f) | !(fabsf(v1[r]) <= 65504.f);
          }
          uint2 h0, h1;
          split4_f8(v0, h0, lo8[2 * half]);
          split4_f8(v1, h1, lo8[2 * half + 1]);
          store_stream16(p.o_fp + ((q_rb + qf) * kbn + head * 2 + half) * 512 + lane * 8, make_uint4(h0.x, h0.y, h1.x, h1.y));
        }
        store_stream16(p.o_lo8 + ((q_rb + qf) * (size_t)p.n_heads + head) * 512 + lane * 8, make_uint4(lo8[0], lo8[1], lo8[2], lo8[3]));
        continue;
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const float v0[4] = {oacc[2 * half][qf][0] * inv, oacc[2 * half][qf][1] * inv, oacc[2 * half][qf][2] * inv,
                             oacc[2 * half][qf][3] * inv};
        const float v1[4] = {oacc[2 * half + 1][qf][0] * inv, oacc[2 * half + 1][qf][1] * inv,
                             oacc[2 * half + 1][qf][2] * inv, oacc[2 * half + 1][qf][3] * inv};
        uint2 h0, l0, h1, l1;
        split4x<O_LO, H16>(v0, h0, l0);
        split4x<O_LO, H16>(v1, h1, l1);
        u16* dst = p.o_fp + (((q_rb + qf) * kbn + head * 2 + half) * 2) * 512 + lane * 8;
        store_stream16(dst, make_uint4(h0.x, h0.y, h1.x, h1.y));
        if (O_LO) store_stream16(dst + 512, make_uint4(l0.x, l0.y, l1.x, l1.y));
      }
    }
    if constexpr (OF8 && H16) {
      if (o_bad && p.range_flag != nullptr) atomicOr(p.range_flag, 1);
    }
  }
}

}  // namespace opk
