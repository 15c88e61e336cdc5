// kernels_fe2.hip — feature extraction in two launches (round 4; replaces fe_curv + fe_pick4 + fe_voxel + fe_collect of kernels_fe.hip,
// which stay behind ALEGO_FE_FUSED=0 as the cross-check and as the path of sort_mode 2).  src/laserOdometry.cpp:122-293.
//
//   fe_cand      a7-a8 + candidates: one wavefront per (stream, ring, sector), all of them at once.  The sector's ranges / columns are staged
//                once in LDS; the 11-tap curvature sum (:122-129), the occlusion / parallel-beam marks (:131-159, as ballot masks — no
//                per-point flag array), the thresholds and each point's suppression reach (how far the +-5 marking of :211-234 gets before a
//                column jump stops it) are evaluated per point, and the sector's sharp / flat CANDIDATES — a 16 x 1800 sector has ~190 points,
//                ~32 sharp and ~12 flat candidates — go to a short list (8 B per candidate; it stays in the L2).  Reads range / column /
//                ground (9 B per point); no curvature or flag array is written.
//   fe_pickc     a9: the greedy pick (:189-277) is sequential per ring and only ever asks for the best remaining candidate: one wavefront per
//                FOUR rings of a stream, sixteen lanes (one DPP row) per ring, the sector's candidates in registers, arg-max by four DPP steps,
//                suppression = an index-range test on the registers; writes the picked indices.
//   fe_ring_out  a10 + the clouds: one workgroup per (stream, ring).  pcl::VoxelGrid(0.4) on the ring's less_flat_scan (:288-293) straight
//                from the segmented cloud — the points of a ring are contiguous, the few labelled ones are holes in a bitmap — then the
//                ring writes its part of all four feature clouds, its index lists, ring offsets and bounding boxes in place.  The less_flat
//                offset of a ring needs the voxel counts of the rings below it: every workgroup publishes its count as soon as it is known
//                (one agent-scope store) and reads the counts below it (a workgroup's ring is its per-stream ticket: the lower rings are
//                always running already).  No staging copy of the filtered rings, no collecting pass.
#include <algorithm>
#include <cstdlib>
#include "dev_common.h"
#include "prof.h"

#define FE_MAXH 4096   // largest horizon_scan (as kernels_fe.hip)

// ---------------------------------------------------------------------------------------------------------------------------------
// fe_cand + fe_pickc
// ---------------------------------------------------------------------------------------------------------------------------------
#ifndef FF_G
#define FF_G 4        // rings per picking wavefront (64 / FF_G lanes each): 4 (default) or 8 — eight rings per wavefront issue a quarter fewer instructions, but the
                      // kernel is a latency chain: four rings finish in 166 instead of 292 us alone and the bench gains 1.1 % (424.9 k -> 429.4 k scans/s, same box)
#endif
#define FF_LPR (64 / FF_G)
#define FF_HALO 8     // staged points either side of a sector (11-tap sum: 5, occlusion marks of the neighbours: 6)
#define FF_Q0 64      // staging position of a sector's first point: the sector's 64-point chunks are the ballot masks' chunks
#ifndef FC_NW
#define FC_NW 4       // ring sectors (wavefronts) per fe_cand workgroup
#endif

struct FfLayout {     // dynamic LDS of one fe_cand wavefront, in bytes
  int scap, mch, wave_bytes;      // staged positions (multiple of 64), mask chunks
};
__host__ __device__ inline FfLayout ff_layout(int sector_cap) {
  FfLayout L;
  L.scap = (FF_Q0 + sector_cap + FF_HALO + 63) & ~63;
  L.mch = L.scap / 64 + 1;
  L.wave_bytes = (8 * 4 * L.mch + 4 * L.scap + 2 * L.scap + 7) & ~7;   // masks u64 [4][mch], ranges f32 [scap], columns | ground << 15 u16 [scap]
  return L;
}

// max over the 64 / FF_G lanes of a ring group (quad_perm xor 1, xor 2, row_half_mirror; row_mirror for 16 lanes), result in every lane of the group
DEV_INLINE uint32_t grp8_max_u32(uint32_t v) {
  int x = (int)v, t;
  t = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false); x = (uint32_t)t > (uint32_t)x ? t : x;
  t = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false); x = (uint32_t)t > (uint32_t)x ? t : x;
  t = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false); x = (uint32_t)t > (uint32_t)x ? t : x;
#if FF_G == 4   // 16 lanes per ring: one more step (row_mirror) = row16_max_u32
  t = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false); x = (uint32_t)t > (uint32_t)x ? t : x;
#endif
  return (uint32_t)x;
}
// bits [pos, pos + 64) of the 128-bit value hi:lo
DEV_INLINE unsigned long long ext128(unsigned long long lo, unsigned long long hi, int pos) { return (lo >> pos) | (pos ? hi << (64 - pos) : 0ull); }

DEV_INLINE void ff_sector(const alego_params& P, int S, int E, int j, int& sp, int& ep) {
  const int NSEC = P.n_sectors;
  if (NSEC == 6) {   // the reference's six sectors: divisions by a constant (a handful of instructions instead of ~35 each)
    if (P.sector_formula == 0) { sp = (S * (6 - j) + E * j) / 6; ep = (S * (5 - j) + E * (j + 1)) / 6 - 1; }
    else { const int diff = E - S; sp = S + j * diff / 6; ep = S + (j + 1) * diff / 6 - 1; }
    return;
  }
  if (P.sector_formula == 0) { sp = (S * (NSEC - j) + E * j) / NSEC; ep = (S * (NSEC - 1 - j) + E * (j + 1)) / NSEC - 1; }   // laserOdometry.cpp:177-178
  else { const int diff = E - S; sp = S + j * diff / NSEC; ep = S + (j + 1) * diff / NSEC - 1; }                               // LO.cpp:245-249
}

// The candidate list of one ring sector, in HBM (it stays in the L2 between fe_cand and fe_pickc): `sector_cap` entries (key, pay); the sharp
// candidates fill it from the front, the flat ones from the back — a point is one or the other (not ground / ground), so they never meet.
//   key  sharp: |cd| bits + 1, flat: ~|cd| bits (0 = no candidate; the pick is an arg-MAX for both; curvature = (double)cd^2 orders like |cd|)
//   pay  sharp: (index in sector) << 6 | reach forward << 3 | reach backward; flat: ~ of that (ties: sharp -> larger index, flat -> smaller)
DEV_INLINE uint2* ff_list(const DevCtx& d, int slot, int ring, int j, int sector_cap) {
  return reinterpret_cast<uint2*>(d.st_lfds + ((size_t)slot * d.NS + ring) * d.H) + (size_t)j * sector_cap;   // (the filtered-ring staging of the four-kernel path: 2 H entries per ring)
}
DEV_INLINE int* ff_counts(const DevCtx& d, int slot, int ring) {   // [sector][2]: sharp, flat candidates (the less_flat_scan index list of the four-kernel path)
  return d.st_idx + ((size_t)slot * d.NS + ring) * d.st_stride + d.cap_sharp + d.cap_lsharp + d.cap_flat;
}

// One sector [sp, ep] of one ring, all 64 lanes of the calling wavefront: the ring's ranges / columns are staged once in LDS; the 11-tap
// curvature sum (:122-129), the occlusion / parallel-beam marks (:131-159) as ballot masks — no per-point flag array — the thresholds and each
// point's suppression reach (how far the +-5 marking of :211-234 gets before a column jump stops it) are evaluated per point, and the
// sector's sharp / flat candidates are compacted into `list`.  Two workgroup barriers (matched by callers that skip a sector).
DEV_INLINE void ff_wide(const DevCtx& d, size_t base, int M, int sp, int ep, unsigned char* wl, const FfLayout& L, uint2* list, int sector_cap, int& ns_out, int& nf_out, bool dbg) {
  const int lane = threadIdx.x & 63;
  const alego_params& P = d.P;
  unsigned long long* mA = reinterpret_cast<unsigned long long*>(wl);
  unsigned long long* mB = mA + L.mch, *mC = mB + L.mch, *mJ = mC + L.mch;
  float* sr = reinterpret_cast<float*>(wl + 8 * 4 * L.mch);
  uint16_t* scl = reinterpret_cast<uint16_t*>(wl + 8 * 4 * L.mch + 4 * L.scap);
  const float* rng = d.seg_range + base;
  const int* colv = d.seg_col + base;
  const uint8_t* gndv = d.seg_ground + base;
  const int len = ep - sp + 1;
  const int q0 = FF_Q0 - FF_HALO, q1 = FF_Q0 + len + FF_HALO;   // staged positions [q0, q1): point k sits at q = k - sp + FF_Q0
#pragma unroll 2
  for (int q = q0 + lane; q < q1; q += 64) {
    const int k = sp + q - FF_Q0;
    const bool in = k >= 0 && k < M;
    const unsigned ku = in ? (unsigned)k : 0u;   // (uniform base + unsigned 32-bit index: no 64-bit address arithmetic per lane)
    const float r = rng[ku];
    const int c = colv[ku];
    const uint8_t g = gndv[ku];
    sr[q] = in ? r : 0.f;
    scl[q] = in ? (uint16_t)((c & 0x7fff) | (g ? 0x8000 : 0)) : (uint16_t)0;
  }
  __syncthreads();
  // per-point predicates of markOccludedPoints (:131-159) and the column jumps of the suppression (:214,:226) as one bit per point
  const int nch = (q1 + 63) >> 6;
  for (int ch = 0; ch < nch; ++ch) {
    const int q = ch * 64 + lane, k = sp + q - FF_Q0;
    const bool inr = q > q0 && q < q1 - 1;
    const int qa = min(max(q, q0 + 1), q1 - 2);
    const float r0 = sr[qa], r1 = sr[qa + 1], rm = sr[qa - 1];
    const int c0 = scl[qa] & 0x7fff, c1 = scl[qa + 1] & 0x7fff;
    int cdiff = c0 - c1;
    cdiff = cdiff < 0 ? -cdiff : cdiff;
    bool c1b, c2b;
    double diff1, diff2;
    if (P.occl_f32) {  // LO.cpp:203-204
      c1b = (double)(r0 - r1) > P.occl_depth; c2b = (double)(r1 - r0) > P.occl_depth;
      diff1 = (double)fabsf(rm - r0); diff2 = (double)fabsf(r1 - r0);
    } else {           // laserOdometry.cpp:134-135
      const double d1 = (double)r0, d2 = (double)r1;
      c1b = d1 - d2 > P.occl_depth; c2b = d2 - d1 > P.occl_depth;
      diff1 = fabs((double)rm - d1); diff2 = fabs(d2 - d1);
    }
    const bool okp = inr && k >= 5 && k < M - 5;
    const bool near = cdiff < P.occl_col_diff;
    const bool A = okp && near && c1b;             // marks i-5..i and skips the rest (:142-144)
    const bool B = okp && near && !c1b && c2b;     // marks i+1..i+5 (:148)
    const bool C = okp && !A && diff1 > P.parallel_ratio * (double)r0 && diff2 > P.parallel_ratio * (double)r0;  // (:154-157)
    const bool J = inr && k + 1 < M && cdiff > P.suppress_col_diff;   // |col[k + 1] - col[k]|
    const unsigned long long a = __ballot(A), b = __ballot(B), c = __ballot(C), j = __ballot(J);
    if (lane == 0) { mA[ch] = a; mB[ch] = b; mC[ch] = c; mJ[ch] = j; }
  }
  __syncthreads();
  int ns = 0, nf = 0;
  const int SR = P.suppress_radius;
  for (int it = 0; it * 64 < len; ++it) {
    const int ch = it + 1, q = ch * 64 + lane, loc = it * 64 + lane, k = sp + loc;
    const bool own = loc < len;
    const bool gnd = (scl[q] & 0x8000) != 0;
    // strictly left-to-right f32 sum (:124); built with -ffp-contract=off
    const float cdv = sr[q - 5] + sr[q - 4] + sr[q - 3] + sr[q - 2] + sr[q - 1] - sr[q] * 10 + sr[q + 1] + sr[q + 2] + sr[q + 3] + sr[q + 4] + sr[q + 5];
    const unsigned long long a0 = mA[ch], a1 = ch + 1 < nch ? mA[ch + 1] : 0ull;
    const unsigned long long bp = mB[ch - 1], b0 = mB[ch];
    const unsigned long long j0 = mJ[ch], j1 = ch + 1 < nch ? mJ[ch + 1] : 0ull, jp = mJ[ch - 1];
    const bool anyA = (ext128(a0, a1, lane) & 0x3full) != 0;                                    // A(i'), i' in [i, i+5]
    const bool anyB = (ext128((bp >> 59) | (b0 << 5), b0 >> 59, lane) & 0x1full) != 0;         // B(i'), i' in [i-5, i-1]
    const bool pk = ((mC[ch] >> lane) & 1ull) || anyA || anyB;
    const unsigned f5 = (unsigned)(ext128(j0, j1, lane) & 0x1full);                             // jumps at i .. i+4
    const unsigned w5 = (unsigned)(ext128((jp >> 59) | (j0 << 5), j0 >> 59, lane) & 0x1full);  // jumps at i-5 .. i-1 (bit 4 = i-1)
    const int rf = min(SR, __ffs((int)(f5 | 0x20u)) - 1), rb = min(SR, __clz((int)(w5 << 27)));
    const double ad = (double)fabsf(cdv), curv = ad * ad;                                       // (double)diff_range * diff_range, exact (:125)
    const bool free_ = own && !pk;
    const bool cs = free_ && !gnd && curv > P.edge_thres;
    const bool cf = free_ && gnd && curv < P.surf_thres;
    const uint32_t kb = (uint32_t)d_f2i(fabsf(cdv));
    const uint32_t pay = ((uint32_t)loc << 6) | ((uint32_t)rf << 3) | (uint32_t)rb;
    const unsigned long long ms = __ballot(cs), mf = __ballot(cf), below = (1ull << lane) - 1ull;
    if (cs) list[(unsigned)(ns + (int)__popcll(ms & below))] = make_uint2(kb + 1u, pay);
    if (cf) list[(unsigned)(sector_cap - 1 - (nf + (int)__popcll(mf & below)))] = make_uint2(~kb, ~pay);
    ns += (int)__popcll(ms); nf += (int)__popcll(mf);
    if (dbg && own) { d.cd[base + k] = cdv; d.picked0[base + k] = pk ? 1 : 0; }
  }
  ns_out = ns; nf_out = nf;
}

// one wavefront per (ring, sector) — every sector of every ring of every stream at once, nothing sequential
__global__ void __launch_bounds__(64 * FC_NW) fe_cand(DevCtx d, int sector_cap) {
  // (the wavefront's number through readfirstlane: ring, sector, their bounds and every pointer derived from them are then scalar — the compiler cannot
  // know that threadIdx.x >> 6 is uniform, and computed all of it per lane with 64-bit vector arithmetic)
  const int slot = blockIdx.y + d.slot0, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int NS = d.NS, NSEC = d.P.n_sectors;
  const int item = blockIdx.x * FC_NW + wave, ring = item / NSEC, j = item - ring * NSEC;
  const size_t base = (size_t)slot * d.N;
  const int M = d.scal[slot * SC_COUNT + SC_M];
  extern __shared__ __attribute__((aligned(16))) unsigned char ff_smem[];
  const FfLayout L = ff_layout(sector_cap);
  int sp = 0, ep = -1;
  if (ring < NS) ff_sector(d.P, d.ring_start[slot * NS + ring], d.ring_end[slot * NS + ring], j, sp, ep);
  int ns = 0, nf = 0;
  if (sp < ep) ff_wide(d, base, M, sp, ep, ff_smem + wave * L.wave_bytes, L, ff_list(d, slot, ring, j, sector_cap), sector_cap, ns, nf, d.n_launch == 1);
  else { __syncthreads(); __syncthreads(); }
  if (ring < NS && lane == 0) { int* cc = ff_counts(d, slot, ring); cc[2 * j] = ns; cc[2 * j + 1] = nf; }
}

// what the eight ring groups of the picking wavefront carry from sector to sector
struct FfPick {
  bool act0;            // the ring exists and the sector has at least two points (:181)
  int sp;               // first point of the sector
  int carry;            // largest index marked so far
  int carry_in;         // ... when the sector began: its candidates at or below it are gone (the marks of a pick are contiguous from the
                        // pick, so everything between the sector's start and carry_in is marked; inside a sector the bitmap / registers decide)
  int n_sharp, n_ls, n_flat;
  int* st_sharp; int* st_lsharp; int* st_flat;
};

// label / continuation of the pick that has just been counted (:196-210, :245-251), shared by the two pick loops
template <bool FLAT>
DEV_INLINE void ff_label(const alego_params& P, int picked, int nmax, int& lab, bool& spread, bool& more) {
  if (FLAT) { lab = -1; more = picked < P.n_flat; spread = more; }                       // the n_flat-th pick breaks before the suppression (:248-251)
  else { lab = picked <= P.n_sharp ? 2 : (picked <= P.n_less_sharp ? 1 : 0); spread = lab != 0; more = lab != 0 && picked < nmax; }
}
// a pick's bookkeeping: sharp picks set their reach in the sector's mark bitmap (the flat candidates are tested against it when they are
// loaded), every pick raises `carry`, lane 0 of the group appends the index to the ring's lists
template <bool FLAT>
DEV_INLINE void ff_commit(FfPick& R, bool rem, int lab, int c, int lo, int hi, uint32_t* mark, int mw, int gl) {
  if (!FLAT) {   // the reach as bits of the sector's mark bitmap: at most two words, lanes 0 and 1 of the group
    const int w = (lo >> 5) + (gl & 1), wl = max(lo, 32 * w), wh = min(hi, 32 * w + 31);
    const uint32_t bits = (rem && gl < 2 && wh >= wl) ? (((2u << (wh - 32 * w)) - 1u) & ~((1u << (wl - 32 * w)) - 1u)) : 0u;
    atomicOr(&mark[min(w, mw - 1)], bits);
  }
  if (rem) R.carry = max(R.carry, R.sp + hi);
  if (rem && gl == 0) {
    if (FLAT) R.st_flat[R.n_flat] = R.sp + c;
    else { if (lab == 2) R.st_sharp[R.n_sharp] = R.sp + c; R.st_lsharp[R.n_ls] = R.sp + c; }
  }
  if (FLAT) R.n_flat += rem ? 1 : 0;
  else { R.n_sharp += (rem && lab == 2) ? 1 : 0; R.n_ls += rem ? 1 : 0; }
}
DEV_INLINE bool ff_marked(const uint32_t* mark, int mw, uint32_t loc) { return (mark[min((int)(loc >> 5), mw - 1)] >> (loc & 31u)) & 1u; }

// The greedy pick of one sector for the FF_G rings of the wavefront, candidates in registers: lane gl of a group holds the list entries
// gl, gl + FF_LPR, ... (K of them; lp[t * dir] = entry t).  FLAT = false: sharp / less-sharp (:189-236), FLAT = true: flat (:238-277).  Every pick:
// arg-max of (key, pay) over the group, label by count, then every candidate inside the picked point's reach leaves the registers.  The
// (n_less_sharp + 1)-th sharp pick of the reference (:207-210: marked, not labelled, break) has no effect on any output — it marks a non-ground
// point of its own sector, and only ground points are flat candidates — and is not taken.
template <int K, bool FLAT>
DEV_INLINE void ff_pick_regs(const alego_params& P, FfPick& R, int ncand, const uint2* lp, int dir, uint32_t* mark, int mw, int gl) {
  uint32_t key[K], pay[K];
#pragma unroll
  for (int t = 0; t < K; ++t) {   // unconditional loads from clamped addresses: all K in flight together
    const int idx = gl + FF_LPR * t;
    const bool ok = R.act0 && idx < ncand;
    const uint2 e = lp[(ok ? idx : 0) * dir];
    const uint32_t pv = FLAT ? ~e.y : e.y, loc = pv >> 6;
    const bool gone = R.sp + (int)loc <= R.carry_in || (FLAT && ff_marked(mark, mw, loc));
    key[t] = (ok && !gone) ? e.x : 0u; pay[t] = e.y;
  }
  const int nmax = FLAT ? 0x7fffffff : max(P.n_sharp, P.n_less_sharp);
  int picked = 0;
  bool act = R.act0 && (FLAT || nmax > 0);
  while (true) {
    uint32_t bk = 0u, bp = 0u;
#pragma unroll
    for (int t = 0; t < K; ++t) {   // a lane's entries come in index order: among equal keys the LAST one for sharp (larger index), the FIRST one for flat
      const bool take = FLAT ? key[t] > bk : key[t] >= bk;
      bk = take ? key[t] : bk; bp = take ? pay[t] : bp;
    }
    if (!act) bk = 0u;
    const uint32_t kmax = grp8_max_u32(bk);
    act = act && kmax != 0u;
    if (!__any(act)) break;
    uint32_t pm = grp8_max_u32(bk == kmax ? bp : 0u);
    if (FLAT) pm = ~pm;
    const int c = (int)(pm >> 6), rf = (int)((pm >> 3) & 7u), rb = (int)(pm & 7u);
    picked += act ? 1 : 0;
    int lab;
    bool spread, more;
    ff_label<FLAT>(P, picked, nmax, lab, spread, more);
    const int lo = max(c - (spread ? rb : 0), 0), hi = c + (spread ? rf : 0);
    const bool rem = act && (FLAT || lab != 0);
#pragma unroll
    for (int t = 0; t < K; ++t) {
      const uint32_t loc = (FLAT ? ~pay[t] : pay[t]) >> 6;
      if (rem && loc - (uint32_t)lo <= (uint32_t)(hi - lo)) key[t] = 0u;
    }
    ff_commit<FLAT>(R, rem, lab, c, lo, hi, mark, mw, gl);
    act = act && more;
  }
}

// the same pick with the candidates left in their list: any number of candidates, any sector length
template <bool FLAT>
DEV_INLINE void ff_pick_mem(const alego_params& P, FfPick& R, int ncand, int nwave, uint2* lp, int dir, uint32_t* mark, int mw, int gl) {
  for (int t = gl; t < nwave; t += FF_LPR) {   // what the earlier sectors' picks (and, for flat, this sector's sharp picks) have marked
    if (R.act0 && t < ncand) {
      const uint2 e = lp[t * dir];
      const uint32_t loc = (FLAT ? ~e.y : e.y) >> 6;
      if (R.sp + (int)loc <= R.carry_in || (FLAT && ff_marked(mark, mw, loc))) lp[t * dir].x = 0u;
    }
  }
  const int nmax = FLAT ? 0x7fffffff : max(P.n_sharp, P.n_less_sharp);
  int picked = 0;
  bool act = R.act0 && (FLAT || nmax > 0);
  while (true) {
    uint32_t bk = 0u, bp = 0u;
    for (int t = gl; t < nwave; t += FF_LPR) {
      if (act && t < ncand) {
        const uint2 e = lp[t * dir];
        const bool take = FLAT ? e.x > bk : e.x >= bk;
        bk = take ? e.x : bk; bp = take ? e.y : bp;
      }
    }
    const uint32_t kmax = grp8_max_u32(bk);
    act = act && kmax != 0u;
    if (!__any(act)) break;
    uint32_t pm = grp8_max_u32(bk == kmax ? bp : 0u);
    if (FLAT) pm = ~pm;
    const int c = (int)(pm >> 6), rf = (int)((pm >> 3) & 7u), rb = (int)(pm & 7u);
    picked += act ? 1 : 0;
    int lab;
    bool spread, more;
    ff_label<FLAT>(P, picked, nmax, lab, spread, more);
    const int lo = max(c - (spread ? rb : 0), 0), hi = c + (spread ? rf : 0);
    const bool rem = act && (FLAT || lab != 0);
    for (int t = gl; t < nwave; t += FF_LPR) {
      if (rem && t < ncand) {
        const uint32_t pv = lp[t * dir].y, loc = (FLAT ? ~pv : pv) >> 6;
        if (loc - (uint32_t)lo <= (uint32_t)(hi - lo)) lp[t * dir].x = 0u;
      }
    }
    ff_commit<FLAT>(R, rem, lab, c, lo, hi, mark, mw, gl);
    act = act && more;
  }
}

// One wavefront per FF_G = 4 rings of a stream (16 lanes each), sector by sector in lock-step: the greedy pick only ever looks at the
// candidates fe_cand left — a 16 x 1800 sector has ~190 points and ~32 sharp candidates — in registers; arg-max by four DPP steps, suppression =
// an index-range test on the registers.  BIG: sectors of more than 400 points (16 x 4000: up to ~140 sharp / ~200 flat candidates).
#define FF_MW_MAX 28   // mark words per ring: sectors of up to 768 points (alego_create) + reach
template <bool BIG>
__global__ void __launch_bounds__(64) fe_pickc(DevCtx d, int sector_cap) {
  const int slot = blockIdx.y + d.slot0, ring0 = blockIdx.x * FF_G, lane = threadIdx.x;
  const int g = lane / FF_LPR, gl = lane % FF_LPR;
  const int NS = d.NS;
  const alego_params& P = d.P;
  int* sc = d.scal + slot * SC_COUNT;
  __shared__ uint32_t s_mark[FF_G][FF_MW_MAX];
  const int mw = min((sector_cap + FF_HALO + 31) / 32 + 1, FF_MW_MAX);
  if (blockIdx.x == 0 && lane == 0) { sc[SC_FE_EPOCH] = sc[SC_FE_EPOCH] + 1; sc[SC_FE_TICKET] = 0; }   // (fe_ring_out of this launch tags its ring counts with the epoch and hands out its rings by ticket)
  const int ring = ring0 + g;
  const bool rv = ring < NS;
  const int rc = rv ? ring : 0;
  const int S = rv ? d.ring_start[slot * NS + ring] : 0, E = rv ? d.ring_end[slot * NS + ring] : 0;
  int* st = d.st_idx + ((size_t)slot * NS + rc) * d.st_stride;
  const int* cc = ff_counts(d, slot, rc);
  FfPick R;
  R.act0 = false; R.sp = 0; R.carry = -1; R.carry_in = -1; R.n_sharp = 0; R.n_ls = 0; R.n_flat = 0;
  R.st_sharp = st; R.st_lsharp = st + d.cap_sharp; R.st_flat = st + d.cap_sharp + d.cap_lsharp;
  uint32_t* mk = s_mark[g];
  const int NSEC = P.n_sectors;
  for (int j = 0; j < NSEC; ++j) {
    int sp = 0, ep = -1;
    if (rv) ff_sector(P, S, E, j, sp, ep);
    R.act0 = rv && sp < ep; R.sp = sp; R.carry_in = R.carry;
    const int ns = R.act0 ? cc[2 * j] : 0, nf = R.act0 ? cc[2 * j + 1] : 0;
    uint2* lst = ff_list(d, slot, rc, j, sector_cap);
    for (int w = gl; w < mw; w += FF_LPR) mk[w] = 0u;
    const int nsw = (int)wave_max_u32((uint32_t)ns), nfw = (int)wave_max_u32((uint32_t)nf);
    constexpr int KU = 32 / FF_LPR;   // registers per lane for 32 candidates of a ring
    if (nsw <= (BIG ? 64 : 32)) ff_pick_regs<(BIG ? 2 : 1) * KU, false>(P, R, ns, lst, 1, mk, mw, gl);
    else if (nsw <= (BIG ? 128 : 64)) ff_pick_regs<(BIG ? 4 : 2) * KU, false>(P, R, ns, lst, 1, mk, mw, gl);
    else if (nsw <= (BIG ? 192 : 96) && !d.opt_fe_cand) ff_pick_regs<(BIG ? 6 : 3) * KU, false>(P, R, ns, lst, 1, mk, mw, gl);
    else ff_pick_mem<false>(P, R, ns, nsw, lst, 1, mk, mw, gl);
    __builtin_amdgcn_wave_barrier();   // (LDS is in order per wavefront: the sharp picks' marks are there before the flat candidates are tested against them)
    if (nfw <= 32 && !(d.opt_fe_cand && nfw > 8)) ff_pick_regs<KU, true>(P, R, nf, lst + sector_cap - 1, -1, mk, mw, gl);
    else if (BIG && nfw <= 64 && !d.opt_fe_cand) ff_pick_regs<2 * KU, true>(P, R, nf, lst + sector_cap - 1, -1, mk, mw, gl);
    else ff_pick_mem<true>(P, R, nf, nfw, lst + sector_cap - 1, -1, mk, mw, gl);
    __builtin_amdgcn_wave_barrier();
  }
  if (rv && gl == 0) {
    int* c = d.st_cnt + ((size_t)slot * NS + ring) * 8;
    c[0] = R.n_sharp; c[1] = R.n_ls; c[2] = R.n_flat;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// fe_ring_out
// ---------------------------------------------------------------------------------------------------------------------------------
#ifndef FO_BLOCK
#define FO_BLOCK 256
#endif
#define FO_NB 256          // buckets of the run ordering (as fe_voxel)
#define FO_U 4             // loads kept in flight per thread
#define FO_CAP_PCT 72
#define FO_INVALID 0xFFFFFFFFu
#define FO_END 0xFFFFu
#define FO_SPIN_LIMIT (1 << 22)
#define FO_ALIAS_INTS (6 * ((FE_MAXH + LO_CH - 1) / LO_CH) > 768 ? 6 * ((FE_MAXH + LO_CH - 1) / LO_CH) : 768)   // bucket tables / hole prefix / box corners: 768 with boxes of 32 targets
#define FO_STATIC_LDS (928 + 4 * FO_ALIAS_INTS) // static LDS of fe_ring_out (bitmap, the aliased bucket / box area, reduction scratch, count table), rounded up: 4000
// points of the ring staged in LDS (by position in the ring): the rest is read from the L2 on every pass.  The workgroup stays within 40 KB
// — FOUR rings per CU; one byte more and it is three (fe_voxel's budget)
#ifndef FO_BUDGET_BIG
#define FO_BUDGET_BIG 53248   // LDS budget of a ring too wide for four workgroups per CU anyway (10 H + static > 40 KB: H = 4000): what three per CU leave, i.e. 600 staged points (80 KB / two per CU: 216 k against 224 k scans/s)
#endif
// dynamic LDS of a ring: 12 bytes per column for the run tables, 2 more where they fit ("keyed": the bucket lists of the run ordering hold 32-bit keys instead of
// 16-bit run numbers, in the place of the per-point keys, and the sorted order moves behind the tables),
// the rest of the budget for staged points
#ifndef FO_KEYED
#define FO_KEYED 1
#endif
#ifndef FO_STAGE
#define FO_STAGE 0   // 1: the first `cap` points of the ring wait in LDS between the passes (measured: nothing gained over reading them from the L2 again; every access pays for the choice)
#endif
struct FoLayout { int cap; int keyed; };
__host__ __device__ inline FoLayout fo_layout(int H) {
  const int budget = 12 * H + FO_STATIC_LDS > 40960 ? FO_BUDGET_BIG : 40960;
  FoLayout L;
  L.keyed = FO_KEYED && 14 * H + FO_STATIC_LDS <= budget;
  const int by_pct = (H * FO_CAP_PCT / 100 + 15) & ~15, by_lds = (budget - FO_STATIC_LDS - (L.keyed ? 14 : 12) * H) / 16;
  L.cap = !FO_STAGE || by_lds <= 0 ? 0 : (by_pct < by_lds ? by_pct : (by_lds & ~15));
  return L;
}
static size_t fo_lds_bytes(int H) { const FoLayout L = fo_layout(H); return std::max((size_t)(L.keyed ? 14 : 12) * H + (size_t)16 * L.cap, (size_t)6 * 65 * 4); }

// reductions over the 64 lanes by DPP (quad, half row, row, then the row results passed on: row_bcast15 / row_bcast31): the result is in lane 63
#define FO_DPP_F(x, ctrl, rmask) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), ctrl, rmask, 0xF, false))
DEV_INLINE float wave_min_f32(float x) {
  x = fminf(x, FO_DPP_F(x, 0xB1, 0xF)); x = fminf(x, FO_DPP_F(x, 0x4E, 0xF)); x = fminf(x, FO_DPP_F(x, 0x141, 0xF)); x = fminf(x, FO_DPP_F(x, 0x140, 0xF));
  x = fminf(x, FO_DPP_F(x, 0x142, 0xA)); x = fminf(x, FO_DPP_F(x, 0x143, 0xC));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
DEV_INLINE float wave_max_f32(float x) {
  x = fmaxf(x, FO_DPP_F(x, 0xB1, 0xF)); x = fmaxf(x, FO_DPP_F(x, 0x4E, 0xF)); x = fmaxf(x, FO_DPP_F(x, 0x141, 0xF)); x = fmaxf(x, FO_DPP_F(x, 0x140, 0xF));
  x = fmaxf(x, FO_DPP_F(x, 0x142, 0xA)); x = fmaxf(x, FO_DPP_F(x, 0x143, 0xC));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
DEV_INLINE int wave_sum_i32(int x) {   // (a disabled row keeps `old` = 0: nothing added)
  x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false); x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false); x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false); x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
  return __builtin_amdgcn_readlane(x, 63);
}

// a * b + c on the low 24 bits of a and b, at full rate (a 32-bit multiply takes four times as long; written out because the compiler turns __umul24 of
// values it cannot bound back into mask + 32-bit multiply)
DEV_INLINE uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// order-preserving map float -> u32 (for LDS atomic min / max) and back
DEV_INLINE uint32_t fo_ord(float f) { const uint32_t b = (uint32_t)d_f2i(f); return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
DEV_INLINE float fo_unord(uint32_t u) { return d_i2f((int32_t)(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu))); }

#ifdef ALEGO_TIMING
__device__ long long fo_times[24];
extern "C" void alego_fo_times(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(fo_times), sizeof(long long) * 24); }
#ifndef FO_TICK_RING
#define FO_TICK_RING 8
#endif
#define FO_TICK(k) do { if (threadIdx.x == 0 && ring == FO_TICK_RING && blockIdx.x == 0) fo_times[k] = wall_clock64(); } while (0)
#elif defined(FO_STOP_AFTER)
// development (instruction counts per phase, tools/fo_phase_counts.sh): the ring stops after phase FO_STOP_AFTER — its count is published first so that no ring above it spins
#define FO_TICK(k) do { if ((k) == FO_STOP_AFTER) { if (threadIdx.x == 0) __hip_atomic_store(&d.fe_sync[(size_t)slot * NS + ring], (epoch << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; } } while (0)
#else
#define FO_TICK(k)
#endif

// the workgroup of row NS: the stream's sharp / less_sharp / flat clouds (ring-ascending concatenation of the picks, :199-205,:245), their
// index lists, the less_sharp ring offsets and bounding boxes (everything fe_pickc counted; no ring has to wait for this)
DEV_INLINE void fo_picks_out(const DevCtx& d, int slot, unsigned char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, NS = d.NS;
  const size_t base = (size_t)slot * d.N;
  const size_t fb = (size_t)slot * 2 + cur_in_flight(d, slot);
  const int* allc = d.st_cnt + (size_t)slot * NS * 8;
  const float4* seg = d.seg_lo + base;
  int (*s_off)[65] = reinterpret_cast<int (*)[65]>(smem);   // [0..2]: first pick of every ring in the three clouds, [3]: first less_sharp box
  if (tid < 64) {
    const int r = tid;
    int c[4];
    c[0] = r < NS ? allc[r * 8 + 0] : 0; c[1] = r < NS ? allc[r * 8 + 1] : 0; c[2] = r < NS ? allc[r * 8 + 2] : 0;
    c[3] = (c[1] + LO_CH - 1) / LO_CH;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int incl = c[k];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
      s_off[k][r] = incl - c[k];
      if (r == 63) s_off[k][64] = incl;
    }
  }
  __syncthreads();
  if (tid <= NS) {
    d.ring_off[(fb * 2) * (NS + 1) + tid] = tid < NS ? s_off[1][tid] : s_off[1][64];
    d.ring_boff[(fb * 2) * (NS + 1) + tid] = tid < NS ? s_off[3][tid] : s_off[3][64];
  }
  if (tid < 3) d.feat_cnt[fb * 4 + tid] = s_off[tid][64];
  auto ring_of = [&](int k, int i) -> int {   // largest r with s_off[k][r] <= i (rings beyond NS hold the total: never chosen for i < total)
    int lo = 0, hi = NS - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_off[k][mid] <= i) lo = mid; else hi = mid - 1; }
    return lo;
  };
  const int stoff[3] = {0, d.cap_sharp, d.cap_sharp + d.cap_lsharp};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float4* dst = d.feat[k] + fb * d.fcap[k];
    int* dsti = d.feat_idx[k] + fb * d.fcap[k];
    float4* bx = d.lo_box + (fb * 2 + 1) * d.lo_box_cap * 2;
    const bool boxes = k == F_LSHARP;
    // less_sharp is walked box by box (LO_CH consecutive threads = the up to LO_CH picks of one ring's box), the others pick by pick
    const int nwork = boxes ? s_off[3][64] * LO_CH : s_off[k][64];
    for (int i0 = 0; i0 < nwork; i0 += FO_BLOCK) {
      int i = i0 + tid, r = 0, j = 0, bxi = 0;
      bool v = i < nwork;
      if (boxes) {
        bxi = min(i / LO_CH, s_off[3][64] - 1);
        r = ring_of(3, bxi); j = (bxi - s_off[3][r]) * LO_CH + (tid % LO_CH);
        v = v && j < s_off[k][r + 1] - s_off[k][r];
        i = s_off[k][r] + j;
      } else if (v) { r = ring_of(k, i); j = i - s_off[k][r]; }
      float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v) {
        const int idx = d.st_idx[((size_t)slot * NS + r) * d.st_stride + stoff[k] + j];
        p = seg[idx];
        dsti[i] = idx; dst[i] = p;
      }
      if (boxes) {
        float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
        if (v) { mn[0] = mx[0] = p.x; mn[1] = mx[1] = p.y; mn[2] = mx[2] = p.z; }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int o = LO_CH / 2; o > 0; o >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64)); }
        if ((tid % LO_CH) == 0 && v) {   // (thread 0 of a box holds its first pick: every box has one)
          bx[2 * bxi] = make_float4(mn[0], mn[1], mn[2], __int_as_float(i));
          bx[2 * bxi + 1] = make_float4(mx[0], mx[1], mx[2], __int_as_float(min(LO_CH, s_off[k][r + 1] - i)));
        }
      }
    }
  }
  // cloud_label_ for the single-scan entry points / tests (:196-204,:245): 2 sharp, 1 less sharp, -1 flat, 0 otherwise
  if (d.n_launch == 1) {
    const int M = d.scal[slot * SC_COUNT + SC_M];
    for (int k = tid; k < M; k += FO_BLOCK) d.plabel[base + k] = 0;
    __syncthreads();
    for (int pass = 0; pass < 3; ++pass) {   // less sharp first: a sharp pick is on both lists
      const int k = pass == 0 ? 1 : (pass == 1 ? 0 : 2), lab = pass == 0 ? 1 : (pass == 1 ? 2 : -1);
      for (int i = tid; i < s_off[k][64]; i += FO_BLOCK) {
        const int r = ring_of(k, i);
        d.plabel[base + d.st_idx[((size_t)slot * NS + r) * d.st_stride + stoff[k] + i - s_off[k][r]]] = lab;
      }
      __syncthreads();
    }
  }
}

// grid (streams, NS + 1): row r < NS = ring r of the stream, row NS = fo_picks_out
__global__ void __launch_bounds__(FO_BLOCK) fe_ring_out(DevCtx d) {
  const int slot = blockIdx.x + d.slot0, tid = threadIdx.x, lane = tid & 63;
  const int NS = d.NS, H = d.H;
  extern __shared__ __attribute__((aligned(16))) unsigned char fo_smem[];
  if ((int)blockIdx.x >= d.n_launch) return;   // (ALEGO_FE_PAD8=1 pads the grid's x extent to a multiple of 8, see launch_fe_fused)
  if ((int)blockIdx.y == NS) { fo_picks_out(d, slot, fo_smem); return; }
  int* scv = d.scal + slot * SC_COUNT;
  // Which ring this workgroup takes is decided by a per-stream TICKET, not by blockIdx.y: a ring waits for the voxel counts of the rings below it, and with
  // tickets those rings belong to workgroups that took theirs earlier — they are running, and they in turn wait only for still earlier ones.  No assumption
  // about the order in which the (per-XCD) dispatchers place the workgroups of a launch is left (round 4 relied on "lower blockIdx.y first on the stream's
  // XCD", which three or six stream groups broke: DESIGN.md section 7).  fe_pickc of the same scan reset the counter.
  __shared__ int s_ticket;
  if (tid == 0) s_ticket = atomicAdd(&scv[SC_FE_TICKET], 1);
  __syncthreads();
  const int ring = min(s_ticket, NS - 1);   // (the clamp only matters if somebody launches the kernel without fe_pickc in front of it)
  const size_t base = (size_t)slot * d.N;
  const alego_params& P = d.P;
  const int cur = cur_in_flight(d, slot);
  const size_t fb = (size_t)slot * 2 + cur;
  const unsigned epoch = (unsigned)scv[SC_FE_EPOCH] & 0xFFFFu;
  const int S = d.ring_start[slot * NS + ring], E = d.ring_end[slot * NS + ring];
  const int n_all = min(max(E - S, 0), H);               // the sectors of a ring cover [S, E - 1] (:177-178)
  const int n_ls = d.st_cnt[((size_t)slot * NS + ring) * 8 + 1];
  const int* st_ls = d.st_idx + ((size_t)slot * NS + ring) * d.st_stride + d.cap_sharp;
  const float4* seg = d.seg_lo + base;
  float4* s_pt = reinterpret_cast<float4*>(fo_smem);                            // the ring's points [cap]
  const FoLayout lay = fo_layout(H);
  const int cap = lay.cap;
  unsigned char* fv2 = fo_smem + (size_t)16 * cap;
  uint32_t* s_key = reinterpret_cast<uint32_t*>(fv2);                           // voxel id per point (a hole carries the id of the point before it)   [H]
  uint16_t* s_tmp = reinterpret_cast<uint16_t*>(fv2);                           // (once the runs are found, in the keys' place) valid runs dealt into buckets [H]; then: first point | flags of the run at that place of the order
  uint32_t* k_tmp = reinterpret_cast<uint32_t*>(fv2);                           // (keyed, in the keys' place as well) per place of the bucket lists: (voxel id within the bucket) << 12 | run [H]
  uint16_t* s_order = reinterpret_cast<uint16_t*>(fv2 + (lay.keyed ? 12 : 2) * (size_t)H);   // valid runs sorted by (voxel id, run) [H]
  uint32_t* s_rvid = reinterpret_cast<uint32_t*>(fv2 + 4 * (size_t)H);          // voxel id per run [H]
  uint16_t* s_rstart = reinterpret_cast<uint16_t*>(fv2 + 8 * (size_t)H);        // first point of the run  [H]
  uint16_t* s_nxt = reinterpret_cast<uint16_t*>(fv2 + 10 * (size_t)H);          // per point: the next point of its voxel in summation order (FO_END: none) [H]
  __shared__ uint32_t s_bm[FE_MAXH / 32 + 2];     // holes of less_flat_scan: the ring's less-sharp picks (label > 0, :284) and points of skipped sectors (:181)
  constexpr int FO_MAXBOX = (FE_MAXH + LO_CH - 1) / LO_CH;
  constexpr int FO_ALIAS = FO_ALIAS_INTS;
  static_assert(6 * FO_MAXBOX <= FO_ALIAS && 2 * (FO_NB + 1) <= FO_ALIAS && FE_MAXH / 32 + 1 <= FO_ALIAS, "s_alias holds the bucket tables, the hole prefix or the box corners");
  __shared__ int s_alias[FO_ALIAS];                    // bucket offsets / cursors while the runs are ordered; hole prefix counts of the pass-through; box corners afterwards
  int* s_boff = s_alias; int* s_bcur = s_alias + FO_NB + 1; int* s_wpre = s_alias;
  uint32_t* s_bx = reinterpret_cast<uint32_t*>(s_alias);                        // [6][FO_MAXBOX]: min xyz, max xyz of every box as ordered u32
  __shared__ float s_red[6][FO_BLOCK / 64];
  __shared__ int s_scan[FO_BLOCK / 64], s_cntv[FO_BLOCK / 64];
  __shared__ int s_look[3];
  __shared__ int s_rc[64];                        // heads per (chunk of FO_BLOCK, wavefront): every wavefront scans the 64 counts itself — one barrier for all chunks
  static_assert(FE_MAXH / 64 <= 64 && FO_BLOCK % 64 == 0, "s_rc holds one count per 64 positions of a ring");
  constexpr int NW = FO_BLOCK / 64;
  const int wv = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  auto scan64 = [&](int cnt, int* total) -> int {   // exclusive prefix of one value per lane over the wavefront
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    *total = __builtin_amdgcn_readlane(incl, 63);
    return incl - cnt;
  };
  if (tid < 64) s_rc[tid] = 0;
  FO_TICK(0);
  // the ring's first FO_U x FO_BLOCK points are fetched now, in front of the bitmap's own round trip (the pick list), and stay in registers for the
  // bounding box and for the voxel ids (most rings have no more points than that)
  float4 keep[FO_U];
#pragma unroll
  for (int u = 0; u < FO_U; ++u) keep[u] = (seg + S)[(unsigned)max(min(tid + u * FO_BLOCK, n_all - 1), 0)];
  const int BW = (n_all + 31) / 32;
  for (int w = tid; w <= BW + 1 && w <= FE_MAXH / 32 + 1; w += FO_BLOCK) s_bm[w] = 0u;
  __syncthreads();
  for (int t = tid; t < n_ls; t += FO_BLOCK) {
    const int i = st_ls[t] - S;
    if (i >= 0 && i < n_all) atomicOr(&s_bm[i >> 5], 1u << (i & 31));
  }
  if (tid < P.n_sectors) {
    int sp, ep;
    ff_sector(P, S, E, tid, sp, ep);
    const int i = sp - S;
    if (sp == ep && i >= 0 && i < n_all) atomicOr(&s_bm[i >> 5], 1u << (i & 31));   // a one-point sector is skipped (sp >= ep)
  }
  __syncthreads();
  FO_TICK(1);
  auto hole = [&](int i) -> bool { return (s_bm[i >> 5] >> (i & 31)) & 1u; };
  // the nearest position at or before i (after i) that is not a hole, -1 (>= n_all) if there is none: one look at the bitmap word unless every position of
  // the word on that side is a hole (bits beyond the ring are clear, s_bm has spare words at its end)
  auto prev_kept = [&](int i) -> int {
    uint32_t w = ~s_bm[i >> 5] & (0xFFFFFFFFu >> (31 - (i & 31)));
    int k = i >> 5;
    while (w == 0u && k > 0) { --k; w = ~s_bm[k]; }
    return w ? k * 32 + 31 - __clz((int)w) : -1;
  };
  auto next_kept = [&](int i) -> int {
    const int j = i + 1;
    uint32_t w = ~s_bm[j >> 5] & (0xFFFFFFFFu << (j & 31));
    int k = j >> 5;
    while (w == 0u && k <= BW) { ++k; w = ~s_bm[k]; }
    return w ? k * 32 + __ffs((int)w) - 1 : n_all;
  };
  const float4* seg_ring = seg + S;   // (uniform base + 32-bit index: no 64-bit address arithmetic per access)
#if FO_STAGE
  auto point = [&](int i) -> float4 { if (i < cap) return s_pt[i]; return seg_ring[(unsigned)i]; };
#else
  auto point = [&](int i) -> float4 { return seg_ring[(unsigned)i]; };
#endif
  // ---- pcl::VoxelGrid on the ring's less_flat_scan: runs of consecutive equal voxel ids, ordered through monotone buckets,
  // centroid in original order — fe_voxel's algorithm on positions of the ring instead of an index list; a hole stays inside the run of the
  // point before it and is left out of the chain the run is summed along.
  // The bounding box is taken over ALL points of the ring, holes included: which points share a voxel (floor(p / leaf)) and the order of the voxels
  // (lexicographic in z, y, x) do not depend on it, only whether the linear ids fit — if they do for this box they do for the smaller one of the points
  // that are left; if not, the exact box decides (bbox_exact: "leaf size too small" returns the input unchanged).
  const float inv = 1.0f / P.less_flat_leaf;
  float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  auto wg_box = [&](int nv) -> int {   // workgroup-wide min / max / sum; ends with every thread holding the results
    nv = wave_sum_i32(nv);
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = wave_min_f32(mn[a]); mx[a] = wave_max_f32(mx[a]); }
    __syncthreads();   // (s_red / s_cntv may still be read from an earlier call)
    if (lane == 0) {
      s_cntv[wv] = nv;
#pragma unroll
      for (int a = 0; a < 3; ++a) { s_red[a][wv] = mn[a]; s_red[3 + a][wv] = mx[a]; }
    }
    __syncthreads();
    nv = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) nv += s_cntv[w];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      mn[a] = s_red[a][0]; mx[a] = s_red[3 + a][0];
#pragma unroll
      for (int w = 1; w < NW; ++w) { mn[a] = fminf(mn[a], s_red[a][w]); mx[a] = fmaxf(mx[a], s_red[3 + a][w]); }
    }
    return nv;
  };
  auto bbox_exact = [&]() {
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = 3.402823466e+38f; mx[a] = -3.402823466e+38f; }
#pragma nounroll
    for (int i = tid; i < n_all; i += FO_BLOCK) {
      if (hole(i)) continue;
      const float4 q = point(i);
      mn[0] = fminf(mn[0], q.x); mn[1] = fminf(mn[1], q.y); mn[2] = fminf(mn[2], q.z);
      mx[0] = fmaxf(mx[0], q.x); mx[1] = fmaxf(mx[1], q.y); mx[2] = fmaxf(mx[2], q.z);
    }
    (void)wg_box(0);
  };
  int nval = 0;
  {
    int nh = 0;
    for (int w = tid; w <= BW; w += FO_BLOCK) nh += __popc(s_bm[w]);   // (bits are set for positions of the ring only)
    for (int i0 = tid; i0 < n_all; i0 += FO_BLOCK * FO_U) {
      float4 pt[FO_U];
#pragma unroll
      for (int u = 0; u < FO_U; ++u) pt[u] = i0 == tid ? keep[u] : seg_ring[(unsigned)min(i0 + u * FO_BLOCK, n_all - 1)];   // (clamped: a point taken twice changes no minimum)
#pragma unroll
      for (int u = 0; u < FO_U; ++u) {
#if FO_STAGE
        const int i = i0 + u * FO_BLOCK;
        if (i < min(n_all, cap)) s_pt[i] = pt[u];
#endif
        mn[0] = fminf(mn[0], pt[u].x); mn[1] = fminf(mn[1], pt[u].y); mn[2] = fminf(mn[2], pt[u].z);
        mx[0] = fmaxf(mx[0], pt[u].x); mx[1] = fmaxf(mx[1], pt[u].y); mx[2] = fmaxf(mx[2], pt[u].z);
      }
    }
    nval = n_all - wg_box(nh);
  }
  FO_TICK(2);
  bool passthrough = false;
  if (nval > 0) {
    auto too_many = [&]() -> bool {
      const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1, dz = (long long)((mx[2] - mn[2]) * inv) + 1;
      return dx * dy * dz > 2147483647LL;
    };
    if (too_many()) { bbox_exact(); passthrough = too_many(); }   // "leaf size too small": the input is returned unchanged
  }
  int nout = 0, nrv = 0, nruns = 0, vex = 0;   // vex: voxels before every (chunk, wavefront) of the order, one per lane
  if (passthrough) {
    if (tid == 0) { int acc = 0; for (int w = 0; w <= BW; ++w) { s_wpre[w] = acc; acc += __popc(s_bm[w]); } }
    nout = nval;
  } else if (nval > 0) {
    int minb[3], divb[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      minb[a] = (int)floorf(mn[a] * inv);
      divb[a] = (int)floorf(mx[a] * inv) - minb[a] + 1;
    }
    const int mul1 = divb[0], mul2 = divb[0] * divb[1];
    // a hole takes the id of the nearest non-hole before it (none: invalid, a leading run that is never ordered), so that the few
    // labelled points do not cut the runs
    // (24-bit multiplies run at full rate, 32-bit ones at a quarter)
    const bool u24 = (unsigned)divb[0] < (1u << 24) && (unsigned)divb[1] < (1u << 24) && (unsigned)divb[2] < (1u << 24) && (unsigned)mul2 < (1u << 24);
    // runs of consecutive equal voxel ids: the heads inside a wavefront are counted per (chunk, wavefront) while the keys are written (the key of the
    // lane before: DPP wave_shr); the first position of a wavefront is compared after the barrier, when every wavefront scans the counts
    auto key_chunk = [&](int c, float4 own) {
      const int i = c * FO_BLOCK + tid;
      uint32_t key = FO_INVALID;
      if (i < n_all) {
        const int src = prev_kept(i);
        float4 q = own;
        if (src != i) q = point(max(src, 0));   // (a hole: the point it inherits from)
        const int i0 = (int)(floorf(q.x * inv) - (float)minb[0]);
        const int i1 = (int)(floorf(q.y * inv) - (float)minb[1]);
        const int i2 = (int)(floorf(q.z * inv) - (float)minb[2]);
        const uint32_t id = u24 ? mad_u24((uint32_t)i2, (uint32_t)mul2, mad_u24((uint32_t)i1, (uint32_t)mul1, (uint32_t)i0)) : (uint32_t)(i0 + i1 * mul1 + i2 * mul2);
        key = src < 0 ? FO_INVALID : id;
        s_key[i] = key;
      }
      const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)key, (int)key, 0x138, 0xF, 0xF, false);   // wave_shr:1 (lane 0 keeps its own: no head counted here)
      const unsigned long long m = __ballot(i < n_all && key != prev);
      if (lane == 0) s_rc[c * NW + wv] = (int)__popcll(m);
    };
#pragma unroll
    for (int c = 0; c < FO_U; ++c) if (c * FO_BLOCK < n_all) key_chunk(c, keep[c]);
#pragma nounroll
    for (int c = FO_U; c * FO_BLOCK < n_all; ++c) key_chunk(c, seg_ring[(unsigned)min(c * FO_BLOCK + tid, n_all - 1)]);
    __syncthreads();
    {
      const int p0 = lane * 64;   // (count `lane` belongs to the 64 positions from p0)
      const int ex = scan64(s_rc[lane] + ((p0 < n_all && (p0 == 0 || s_key[p0] != s_key[p0 - 1])) ? 1 : 0), &nruns);
#pragma nounroll
      for (int c = 0; c * FO_BLOCK < n_all; ++c) {
        const int i = c * FO_BLOCK + tid;
        const uint32_t mykey = i < n_all ? s_key[i] : 0u;
        const bool head = i < n_all && (i == 0 || mykey != s_key[i - 1]);
        const unsigned long long m = __ballot(head);
        const int r0 = __builtin_amdgcn_readlane(ex, __builtin_amdgcn_readfirstlane(c * NW + wv));
        if (head) {
          const int r = r0 + (int)__popcll(m & lt);
          s_rvid[r] = mykey;
          s_rstart[r] = (uint16_t)i;
        }
        // the chain a voxel's sum follows: the next point of the run that is not a hole (a hole carries the key of the point before it, so it never ends a run)
        if (i < n_all) {
          const int j = next_kept(i);
          s_nxt[i] = (uint16_t)((j < n_all && s_key[min(j, n_all - 1)] == mykey) ? j : FO_END);
        }
      }
    }
    __syncthreads();   // (the keys are dead from here on: the bucket lists take their place)
    FO_TICK(3);
    // order the valid runs by (voxel id, run index): dealt into <= FO_NB buckets monotone in the voxel id, ranked inside the bucket
    {
      unsigned T = (unsigned)divb[0] * (unsigned)divb[1] * (unsigned)divb[2];
      if (T == 0) T = 1;
      int shift = 0;
      while (((T - 1) >> shift) >= (unsigned)FO_NB) ++shift;
      const int nb = (int)((T - 1) >> shift) + 1;
      for (int b = tid; b <= nb; b += FO_BLOCK) s_boff[b] = 0;
      __syncthreads();
      FO_TICK(9);
      for (int r = tid; r < nruns; r += FO_BLOCK) { const uint32_t v = s_rvid[r]; if (v != FO_INVALID) atomicAdd(&s_boff[min((int)(v >> shift), nb - 1) + 1], 1); }
      __syncthreads();
      FO_TICK(10);
      {
        constexpr int PER = FO_NB / FO_BLOCK;
        int v[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) { const int b = tid * PER + k; v[k] = b < nb ? s_boff[b + 1] : 0; sum += v[k]; }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) s_scan[tid >> 6] = incl;
        __syncthreads();
        int run = incl - sum;
#pragma unroll
        for (int w = 0; w < FO_BLOCK / 64; ++w) if (w < (tid >> 6)) run += s_scan[w];
#pragma unroll
        for (int k = 0; k < PER; ++k) { const int b = tid * PER + k; run += v[k]; if (b < nb) { s_boff[b + 1] = run; s_bcur[b + 1] = run; } }
        if (tid == 0) s_bcur[0] = 0;
      }
      __syncthreads();
      nrv = s_boff[nb];
      FO_TICK(11);
      const bool keyed = lay.keyed && shift <= 20;   // (voxel id within a bucket: `shift` bits, run: 12)
      const uint32_t lowmask = (1u << shift) - 1u;
      for (int r = tid; r < nruns; r += FO_BLOCK) {
        const uint32_t v = s_rvid[r];
        if (v != FO_INVALID) {
          const int pos = atomicAdd(&s_bcur[min((int)(v >> shift), nb - 1)], 1);   // s_bcur[b] starts at s_boff[b] (written one slot up, read one down)
          if (keyed) k_tmp[pos] = ((v & lowmask) << 12) | (uint32_t)r;
          else s_tmp[pos] = (uint16_t)r;
        }
      }
      __syncthreads();
      FO_TICK(12);
#ifdef ALEGO_TIMING
      if (tid == 0 && ring == FO_TICK_RING && blockIdx.x == 0) { int mxb = 0; for (int b = 0; b < nb; ++b) mxb = max(mxb, s_boff[b + 1] - s_boff[b]); fo_times[16] = n_all; fo_times[17] = nruns; fo_times[18] = nrv; fo_times[19] = nb; fo_times[20] = mxb; }
#endif
      for (int t = tid; t < nrv; t += FO_BLOCK) {
        const uint32_t mine = keyed ? k_tmp[t] : 0u;
        const int r = keyed ? (int)(mine & 0xFFFu) : (int)s_tmp[t];
        const uint32_t v = s_rvid[r];
        const int b = min((int)(v >> shift), nb - 1);
        const int bs = s_boff[b], be = s_boff[b + 1];
        int rank = bs;
        if (keyed) {   // one read and one compare per entry of the bucket: (id, run) order = order of the packed keys
#pragma nounroll
          for (int q = bs; q < be; ++q) rank += k_tmp[q] < mine ? 1 : 0;
        } else {
#pragma nounroll
          for (int q = bs; q < be; ++q) { const int o = s_tmp[q]; const uint32_t u = s_rvid[o]; rank += (u < v) || (u == v && o < r); }
        }
        s_order[rank] = (uint16_t)r;
      }
    }
    if (tid < 64) s_rc[tid] = 0;   // (last read before the barrier that ended the runs)
    __syncthreads();
    FO_TICK(4);
    // the first point of the run at every place of the order, with a mark where the run continues the voxel of the place before — there the chain of
    // that voxel's previous run is linked to this one; voxels = places that do not continue: counted per (chunk, wavefront) like the runs
#pragma nounroll
    for (int c = 0; c * FO_BLOCK < nrv; ++c) {
      const int j = c * FO_BLOCK + tid;
      uint32_t st = 0x1000u;
      if (j < nrv) {
        const int r = s_order[j];
        const int i0 = s_rstart[r];
        st = (uint32_t)i0;
        if (j > 0) {
          const int rp = s_order[j - 1];
          if (s_rvid[rp] == s_rvid[r]) {
            int t = (rp + 1 < nruns ? (int)s_rstart[rp + 1] : n_all) - 1;
            while (hole(t)) --t;   // (a run's first point is never a hole)
            s_nxt[t] = (uint16_t)i0;
            st |= 0x1000u;
          }
        }
        s_tmp[j] = (uint16_t)st;
      }
      const unsigned long long m = __ballot(!(st & 0x1000u));
      if (lane == 0) s_rc[c * NW + wv] = (int)__popcll(m);
    }
    __syncthreads();
    vex = scan64(s_rc[lane], &nout);
  }
  FO_TICK(5);
  // ---- the ring's less_flat offset: its count for the rings above, the counts of the rings below
  if (tid == 0) __hip_atomic_store(&d.fe_sync[(size_t)slot * NS + ring], (epoch << 16) | (unsigned)nout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid < 64) {
    int c = 0, nbx = 0, bad = 0;
    if (tid < ring) {
      unsigned v = 0;
      int spins = 0;
      const int limit = d.opt_fo_spin > 0 ? d.opt_fo_spin : FO_SPIN_LIMIT;
      while (true) {
        v = __hip_atomic_load(&d.fe_sync[(size_t)slot * NS + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 16) == epoch || ++spins > limit) break;
        __builtin_amdgcn_s_sleep(2);
      }
      bad = (v >> 16) != epoch;
      c = bad ? 0 : (int)(v & 0xFFFFu);   // (a count that never came counts as 0: every offset below only shrinks, so whatever this ring still writes stays inside the slot's arrays)
      nbx = (c + LO_CH - 1) / LO_CH;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o, 64); nbx += __shfl_xor(nbx, o, 64); bad += __shfl_xor(bad, o, 64); }
    if (lane == 0) { s_look[0] = c; s_look[1] = nbx; s_look[2] = bad; }
  }
  const int nbox = (nout + LO_CH - 1) / LO_CH;
  __syncthreads();   // (also: the bucket tables / the hole prefix of the pass-through are dead or read-only from here on... see below)
  const int off3 = s_look[0], boff3 = s_look[1];
  // Gave up on a lower ring (never expected: its workgroup holds an earlier ticket and is running — the limit is seconds).  The flag is sticky and
  // makes every reader of the slot's less_flat cloud treat it as empty (lo_grid_build, lo_assoc, lm_stage) until the host, which gets ALEGO_ERR_HIP from
  // fetch_pose, resets the slot; this ring still writes its (meaningless, in-bounds) part so that the tables of the scan are all written.
  if (s_look[2] && tid == 0) scv[SC_FE_ERR] = 1;
  FO_TICK(6);
  float4* out = d.feat[F_LFLAT] + fb * d.fcap[F_LFLAT] + off3;
  // box corners of every LO_CH consecutive output points, gathered with LDS atomics while the points are written
  auto box_add = [&](int rank, const float4& p) {
    uint32_t* b = s_bx + rank / LO_CH;
    const uint32_t ox = fo_ord(p.x), oy = fo_ord(p.y), oz = fo_ord(p.z);
    atomicMin(&b[0], ox); atomicMin(&b[FO_MAXBOX], oy); atomicMin(&b[2 * FO_MAXBOX], oz);
    atomicMax(&b[3 * FO_MAXBOX], ox); atomicMax(&b[4 * FO_MAXBOX], oy); atomicMax(&b[5 * FO_MAXBOX], oz);
  };
  if (passthrough) {   // (rare; the hole prefix shares its LDS with the box corners: the boxes are taken from the written points afterwards)
    for (int i = tid; i < n_all; i += FO_BLOCK)
      if (!hole(i)) out[i - (s_wpre[i >> 5] + __popc(s_bm[i >> 5] & ((1u << (i & 31)) - 1u)))] = point(i);
    __syncthreads();
    for (int bi = tid; bi < nbox; bi += FO_BLOCK) {
      uint32_t* c = s_bx + bi;
      float lo3[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, hi3[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
      for (int t = bi * LO_CH; t < min(nout, bi * LO_CH + LO_CH); ++t) {
        const float4 q = out[t];
        lo3[0] = fminf(lo3[0], q.x); lo3[1] = fminf(lo3[1], q.y); lo3[2] = fminf(lo3[2], q.z);
        hi3[0] = fmaxf(hi3[0], q.x); hi3[1] = fmaxf(hi3[1], q.y); hi3[2] = fmaxf(hi3[2], q.z);
      }
      __builtin_amdgcn_s_waitcnt(0);   // (all of the thread's reads of s_wpre's area are long done: the barrier above)
      c[0] = fo_ord(lo3[0]); c[FO_MAXBOX] = fo_ord(lo3[1]); c[2 * FO_MAXBOX] = fo_ord(lo3[2]);
      c[3 * FO_MAXBOX] = fo_ord(hi3[0]); c[4 * FO_MAXBOX] = fo_ord(hi3[1]); c[5 * FO_MAXBOX] = fo_ord(hi3[2]);
    }
  } else {
    for (int b = tid; b < 6 * FO_MAXBOX; b += FO_BLOCK) s_bx[b] = b < 3 * FO_MAXBOX ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    FO_TICK(13);
    // first run of every voxel -> output rank; it follows the voxel's chain through all its runs, one point per step
#pragma nounroll
    for (int c = 0; c * FO_BLOCK < nrv; ++c) {
      const int j = c * FO_BLOCK + tid;
      const uint32_t a0 = j < nrv ? s_tmp[j] : 0x1000u;
      const bool head = !(a0 & 0x1000u);
      const unsigned long long m = __ballot(head);
      const int rank0 = __builtin_amdgcn_readlane(vex, __builtin_amdgcn_readfirstlane(c * NW + wv));
      if (head) {
        const int rank = rank0 + (int)__popcll(m & lt);
        int i = (int)(a0 & 0xFFFu);
        float4 q = point(i);
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        int cnt = 0;
#pragma nounroll
        for (;;) {
          const int ni = s_nxt[i];
          sx += q.x; sy += q.y; sz += q.z; si += q.w; ++cnt;   // strictly in order
          if (ni == FO_END) break;
          q = point(ni); i = ni;
        }
        const float fn = (float)cnt;
        const float4 ctr = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
        out[rank] = ctr;
        box_add(rank, ctr);
      }
    }
  }
  __syncthreads();
  FO_TICK(7);
  {
    float4* bx = d.lo_box + (fb * 2 + 0) * d.lo_box_cap * 2 + (size_t)2 * boff3;
    for (int b = tid; b < nbox; b += FO_BLOCK) {
      const uint32_t* c = s_bx + b;
      bx[2 * b] = make_float4(fo_unord(c[0]), fo_unord(c[FO_MAXBOX]), fo_unord(c[2 * FO_MAXBOX]), __int_as_float(off3 + b * LO_CH));
      bx[2 * b + 1] = make_float4(fo_unord(c[3 * FO_MAXBOX]), fo_unord(c[4 * FO_MAXBOX]), fo_unord(c[5 * FO_MAXBOX]), __int_as_float(min(LO_CH, nout - b * LO_CH)));
    }
    if (tid == 0) {
      int* ro = d.ring_off + (fb * 2 + 1) * (NS + 1);
      int* rb = d.ring_boff + (fb * 2 + 1) * (NS + 1);
      ro[ring] = off3; rb[ring] = boff3;
      if (ring == NS - 1) { ro[NS] = off3 + nout; rb[NS] = boff3 + nbox; d.feat_cnt[fb * 4 + 3] = off3 + nout; }
      d.st_cnt[((size_t)slot * NS + ring) * 8 + 4] = nout;
    }
  }
  FO_TICK(8);
}

void launch_fe_curv_debug(const DevCtx& d, hipStream_t st);   // kernels_fe.hip: fe_curv alone (curvature sums / occlusion marks of the points outside every sector, tests only)

// fused path: everything but alego_params.sort_mode = 2 (the libstdc++ tie order needs the whole sector's keys: four-kernel path)
// (and only sector counts whose candidate lists fit where they are kept: n_sectors * sector_cap 8-byte entries in a ring's 16 H-byte staging row (ff_list),
// 2 n_sectors counts in the H-int tail of its index row (ff_counts), one thread per sector for the one-point sectors (fe_ring_out) — anything else, e.g.
// n_sectors > H / 3, takes the four-kernel path)
bool fe_fused_eligible(const DevCtx& d) {
  const int nsec = d.P.n_sectors;
  if (!(d.opt_fe_fused && !d.opt_fe_pick1 && d.P.sort_mode != 2 && d.NS <= 64 && d.H <= FE_MAXH && nsec >= 1)) return false;
  const long long sector_cap = (d.H + nsec - 1) / nsec + 2;
  return nsec <= FO_BLOCK && (long long)nsec * sector_cap <= 2LL * d.H && 2LL * nsec <= d.H;
}

void launch_fe_fused(const DevCtx& d, hipStream_t st) {
  static const bool cfg = hipFuncSetAttribute(reinterpret_cast<const void*>(fe_ring_out), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fo_lds_bytes(FE_MAXH)) == hipSuccess;
  (void)cfg;
  const int sector_cap = (d.H + d.P.n_sectors - 1) / (d.P.n_sectors > 0 ? d.P.n_sectors : 1) + 2;
  if (d.n_launch == 1) launch_fe_curv_debug(d, st);
  const FfLayout L = ff_layout(sector_cap);
  ALEGO_LAUNCH(fe_cand, dim3((d.NS * d.P.n_sectors + FC_NW - 1) / FC_NW, d.n_launch), dim3(64 * FC_NW), (size_t)FC_NW * L.wave_bytes, st, d, sector_cap);
  // registers hold up to 96 sharp + 32 flat candidates of a ring sector (192 + 64 for sectors of more than 400 points: 16 x 4000 has up to ~140 /
  // ~200); longer lists are picked from memory (ff_pick_mem; ALEGO_FE_CAND != 0 sends everything beyond 64 / 8 there: tests)
  const dim3 gp((d.NS + FF_G - 1) / FF_G, d.n_launch);
  if (sector_cap > 400) { ALEGO_LAUNCH(fe_pickc<true>, gp, dim3(64), 0, st, d, sector_cap); }
  else { ALEGO_LAUNCH(fe_pickc<false>, gp, dim3(64), 0, st, d, sector_cap); }
  // Rings are handed out by ticket (fe_ring_out), so the look-back between the rings of a stream does not depend on where the dispatchers put the workgroups.
  // Round 4 padded x to a multiple of 8 instead (workgroup b runs on XCD b mod 8: a stream's rings then share an XCD and its dispatcher's order) after 683
  // streams per launch — three stream groups — had died with a memory access fault; ALEGO_FE_PAD8=1 still does that (a development switch, not needed).
  ALEGO_LAUNCH(fe_ring_out, dim3(d.opt_fo_pad8 ? (d.n_launch + 7) / 8 * 8 : d.n_launch, d.NS + 1), dim3(FO_BLOCK), fo_lds_bytes(d.H), st, d);
}
