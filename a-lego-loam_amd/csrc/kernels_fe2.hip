// kernels_fe2.hip — feature extraction in two launches (round 4; replaces fe_curv + fe_pick4 + fe_voxel + fe_collect of kernels_fe.hip,
// which stay behind ALEGO_FE_FUSED=0 as the cross-check and as the path of sort_mode 2).  src/laserOdometry.cpp:122-293.
//
//   fe_front     a7-a9: one wavefront group per EIGHT rings of a stream walks the rings sector by sector.  For a sector the ring's ranges /
//                columns are staged once in LDS; the 11-tap curvature sum (:122-129), the occlusion / parallel-beam marks (:131-159, as
//                ballot masks — no per-point flag array), the thresholds and each point's suppression reach (how far the +-5 marking of
//                :211-234 gets before a column jump stops it) are evaluated with all 64 lanes, and the sector's sharp / flat CANDIDATES
//                are compacted into a list in LDS.  The greedy pick (:189-277) then only ever looks at candidates — a 16 x 1800 sector has
//                ~190 points and ~32 sharp candidates — in registers, eight rings in lock-step, eight lanes per ring: arg-max by three DPP
//                steps, suppression = an index-range test on the registers.  Nothing per point goes to HBM: the kernel reads range /
//                column / ground (9 B per point) and writes the picked indices.
//   fe_ring_out  a10 + the clouds: one workgroup per (stream, ring).  pcl::VoxelGrid(0.4) on the ring's less_flat_scan (:288-293) straight
//                from the segmented cloud — the points of a ring are contiguous, the few labelled ones are holes in a bitmap — then the
//                ring writes its part of all four feature clouds, its index lists, ring offsets and bounding boxes in place.  The less_flat
//                offset of a ring needs the voxel counts of the rings below it: every workgroup publishes its count as soon as it is known
//                (one agent-scope store) and reads the counts below it (workgroups of lower rings are dispatched earlier: slot-fastest
//                grid).  No staging copy of the filtered rings, no collecting pass.
#include <algorithm>
#include <cstdlib>
#include "dev_common.h"
#include "prof.h"

#define FE_MAXH 4096   // largest horizon_scan (as kernels_fe.hip)

// ---------------------------------------------------------------------------------------------------------------------------------
// fe_front
// ---------------------------------------------------------------------------------------------------------------------------------
#define FF_G 8        // rings per workgroup (8 lanes of the picking wavefront each)
#define FF_HALO 8     // staged points either side of a sector (11-tap sum: 5, occlusion marks of the neighbours: 6)
#define FF_Q0 64      // staging position of a sector's first point: the sector's 64-point chunks are the ballot masks' chunks

struct FfLayout {     // dynamic LDS of one workgroup, in bytes from its start
  int scap, mch, ct, mw;          // staged positions per wavefront (multiple of 64), mask chunks, list entries per ring, mark words per ring
  int off_key, off_pay, off_mark, off_misc, off_wave, wave_bytes, total;
};
__host__ __device__ inline FfLayout ff_layout(int sector_cap, int CS, int CF, int NW) {
  FfLayout L;
  L.scap = (FF_Q0 + sector_cap + FF_HALO + 63) & ~63;
  L.mch = L.scap / 64 + 1;
  L.ct = CS + CF;
  L.mw = (sector_cap + FF_HALO + 31) / 32 + 1;
  L.off_key = 0;
  L.off_pay = L.off_key + 4 * FF_G * L.ct;
  L.off_mark = (L.off_pay + 2 * FF_G * L.ct + 3) & ~3;
  L.off_misc = L.off_mark + 4 * FF_G * L.mw;
  L.off_wave = (L.off_misc + 4 * FF_G * 4 + 7) & ~7;
  L.wave_bytes = (8 * 4 * L.mch + 4 * L.scap + 2 * L.scap + 7) & ~7;   // masks u64 [4][mch], ranges f32 [scap], columns u16 [scap]
  L.total = L.off_wave + NW * L.wave_bytes;
  return L;
}

// max over the 8 lanes of a ring group (quad_perm xor 1, xor 2, row_half_mirror), result in every lane of the group
DEV_INLINE uint32_t grp8_max_u32(uint32_t v) {
  int x = (int)v, t;
  t = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false); x = (uint32_t)t > (uint32_t)x ? t : x;
  t = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false); x = (uint32_t)t > (uint32_t)x ? t : x;
  t = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false); x = (uint32_t)t > (uint32_t)x ? t : x;
  return (uint32_t)x;
}
// bits [pos, pos + 64) of the 128-bit value hi:lo
DEV_INLINE unsigned long long ext128(unsigned long long lo, unsigned long long hi, int pos) { return (lo >> pos) | (pos ? hi << (64 - pos) : 0ull); }

// One sector [sp, ep] of one ring, all 64 lanes of the calling wavefront: candidates -> keyl / payl (sharp from entry 0, flat from entry CS;
// entries beyond CS / CF go to the overflow lists in HBM).  carry = largest index already marked by the picks of the ring's previous sector.
//   key  sharp: |cd| bits + 1, flat: ~|cd| bits (0 = no candidate; the pick is an arg-MAX for both; curvature = (double)cd^2 orders like |cd|)
//   pay  sharp: (index in sector) << 6 | reach forward << 3 | reach backward; flat: 0xFFFF - that (ties: sharp -> larger index, flat -> smaller)
DEV_INLINE void ff_wide(const DevCtx& d, size_t base, int M, int sp, int ep, int carry, unsigned char* wl, const FfLayout& L, uint32_t* keyl, uint16_t* payl,
                        int CS, int CF, uint2* ovf_s, uint2* ovf_f, int& ns_out, int& nf_out, bool dbg) {
  const int lane = threadIdx.x & 63;
  const alego_params& P = d.P;
  unsigned long long* mA = reinterpret_cast<unsigned long long*>(wl);
  unsigned long long* mB = mA + L.mch, *mC = mB + L.mch, *mJ = mC + L.mch;
  float* sr = reinterpret_cast<float*>(wl + 8 * 4 * L.mch);
  uint16_t* scl = reinterpret_cast<uint16_t*>(wl + 8 * 4 * L.mch + 4 * L.scap);
  const float* rng = d.seg_range + base;
  const int* colv = d.seg_col + base;
  const int len = ep - sp + 1;
  const int q0 = FF_Q0 - FF_HALO, q1 = FF_Q0 + len + FF_HALO;   // staged positions [q0, q1): point k sits at q = k - sp + FF_Q0
#pragma unroll 2
  for (int q = q0 + lane; q < q1; q += 64) {
    const int k = sp + q - FF_Q0;
    const bool in = k >= 0 && k < M;
    const float r = rng[in ? k : 0];
    const int c = colv[in ? k : 0];
    sr[q] = in ? r : 0.f;
    scl[q] = in ? (uint16_t)c : (uint16_t)0;
  }
  __syncthreads();
  // per-point predicates of markOccludedPoints (:131-159) and the column jumps of the suppression (:214,:226) as one bit per point
  const int nch = (q1 + 63) >> 6;
  for (int ch = 0; ch < nch; ++ch) {
    const int q = ch * 64 + lane, k = sp + q - FF_Q0;
    const bool inr = q > q0 && q < q1 - 1;
    const int qa = min(max(q, q0 + 1), q1 - 2);
    const float r0 = sr[qa], r1 = sr[qa + 1], rm = sr[qa - 1];
    const int c0 = scl[qa], c1 = scl[qa + 1];
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
    const uint8_t gnd = d.seg_ground[base + (own ? k : sp)];
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
    const bool free_ = own && !pk && k > carry;
    const bool cs = free_ && !gnd && curv > P.edge_thres;
    const bool cf = free_ && gnd && curv < P.surf_thres;
    const uint32_t kb = (uint32_t)d_f2i(fabsf(cdv));
    const uint32_t pay = ((uint32_t)loc << 6) | ((uint32_t)rf << 3) | (uint32_t)rb;
    const unsigned long long ms = __ballot(cs), mf = __ballot(cf), below = (1ull << lane) - 1ull;
    if (cs) {
      const int pos = ns + (int)__popcll(ms & below);
      if (pos < CS) { keyl[pos] = kb + 1u; payl[pos] = (uint16_t)pay; } else ovf_s[pos - CS] = make_uint2(kb + 1u, pay);
    }
    if (cf) {
      const int pos = nf + (int)__popcll(mf & below);
      if (pos < CF) { keyl[CS + pos] = ~kb; payl[CS + pos] = (uint16_t)(0xFFFFu - pay); } else ovf_f[pos - CF] = make_uint2(~kb, 0xFFFFu - pay);
    }
    ns += (int)__popcll(ms); nf += (int)__popcll(mf);
    if (dbg && own) { d.cd[base + k] = cdv; d.picked0[base + k] = pk ? 1 : 0; }
  }
  ns_out = ns; nf_out = nf;
}

// what the eight ring groups of the picking wavefront carry from sector to sector
struct FfPick {
  bool act0;            // the ring exists and the sector has at least two points (:181)
  int sp;               // first point of the sector
  int carry;            // largest index marked so far
  int n_sharp, n_ls, n_flat;
  int* st_sharp; int* st_lsharp; int* st_flat;
};

// The greedy pick of one sector for the eight rings of the wavefront, candidates in registers: lane gl of a group holds the list entries
// gl, gl + 8, ... (K of them).  FLAT = false: sharp / less-sharp (:189-236), FLAT = true: flat (:238-277).  Every pick: arg-max of (key, pay)
// over the group (ties: see ff_wide), label by count, then every candidate inside the picked point's reach leaves the registers; sharp
// picks also set their reach in the sector's mark bitmap (the flat candidates are tested against it when they are loaded) and every pick
// raises `carry`, the marks that reach into the next sector.  The (n_less_sharp + 1)-th sharp pick of the reference (:207-210: marked, not
// labelled, break) has no effect on any output — it marks a non-ground point of its own sector, and only ground points are flat candidates — and
// is not taken.
template <int K, bool FLAT>
DEV_INLINE void ff_pick_regs(const alego_params& P, FfPick& R, int ncand, const uint32_t* keyl, const uint16_t* payl, uint32_t* mark, int mw, int gl) {
  uint32_t key[K], pay[K];
#pragma unroll
  for (int t = 0; t < K; ++t) {
    const int idx = gl + 8 * t;
    const bool ok = R.act0 && idx < ncand;
    const int ia = ok ? idx : 0;
    uint32_t kv = keyl[ia];
    const uint32_t pv = payl[ia];
    if (FLAT) { const uint32_t loc = (0xFFFFu - pv) >> 6; if ((mark[min((int)(loc >> 5), mw - 1)] >> (loc & 31u)) & 1u) kv = 0u; }
    key[t] = ok ? kv : 0u; pay[t] = pv;
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
    if (FLAT) pm = 0xFFFFu - pm;
    const int c = (int)(pm >> 6), rf = (int)((pm >> 3) & 7u), rb = (int)(pm & 7u);
    picked += act ? 1 : 0;
    int lab;
    bool spread, more;
    if (FLAT) { lab = -1; more = picked < P.n_flat; spread = more; }                       // the n_flat-th pick breaks before the suppression (:248-251)
    else { lab = picked <= P.n_sharp ? 2 : (picked <= P.n_less_sharp ? 1 : 0); spread = lab != 0; more = lab != 0 && picked < nmax; }
    const int lo = max(c - (spread ? rb : 0), 0), hi = c + (spread ? rf : 0);
    const bool rem = act && (FLAT || lab != 0);
#pragma unroll
    for (int t = 0; t < K; ++t) {
      const uint32_t loc = (FLAT ? 0xFFFFu - pay[t] : pay[t]) >> 6;
      if (rem && loc - (uint32_t)lo <= (uint32_t)(hi - lo)) key[t] = 0u;
    }
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
    act = act && more;
  }
}

// the same pick with the candidates left where ff_wide put them (LDS, overflow in HBM): any number of candidates, any sector length
template <bool FLAT>
DEV_INLINE void ff_pick_mem(const alego_params& P, FfPick& R, int ncand, int nwave, uint32_t* keyl, const uint16_t* payl, int cap, uint2* ovf, uint32_t* mark, int mw, int gl) {
  auto load = [&](int t, uint32_t& kv, uint32_t& pv) {
    if (t < cap) { kv = keyl[t]; pv = payl[t]; } else { const uint2 e = ovf[t - cap]; kv = e.x; pv = e.y; }
  };
  auto kill = [&](int t) { if (t < cap) keyl[t] = 0u; else ovf[t - cap].x = 0u; };
  if (FLAT) {
    for (int t = gl; t < nwave; t += 8) {
      if (R.act0 && t < ncand) {
        uint32_t kv, pv;
        load(t, kv, pv);
        const uint32_t loc = (0xFFFFu - pv) >> 6;
        if ((mark[min((int)(loc >> 5), mw - 1)] >> (loc & 31u)) & 1u) kill(t);
      }
    }
  }
  const int nmax = FLAT ? 0x7fffffff : max(P.n_sharp, P.n_less_sharp);
  int picked = 0;
  bool act = R.act0 && (FLAT || nmax > 0);
  while (true) {
    uint32_t bk = 0u, bp = 0u;
    for (int t = gl; t < nwave; t += 8) {
      if (act && t < ncand) {
        uint32_t kv, pv;
        load(t, kv, pv);
        const bool take = FLAT ? kv > bk : kv >= bk;
        bk = take ? kv : bk; bp = take ? pv : bp;
      }
    }
    const uint32_t kmax = grp8_max_u32(bk);
    act = act && kmax != 0u;
    if (!__any(act)) break;
    uint32_t pm = grp8_max_u32(bk == kmax ? bp : 0u);
    if (FLAT) pm = 0xFFFFu - pm;
    const int c = (int)(pm >> 6), rf = (int)((pm >> 3) & 7u), rb = (int)(pm & 7u);
    picked += act ? 1 : 0;
    int lab;
    bool spread, more;
    if (FLAT) { lab = -1; more = picked < P.n_flat; spread = more; }
    else { lab = picked <= P.n_sharp ? 2 : (picked <= P.n_less_sharp ? 1 : 0); spread = lab != 0; more = lab != 0 && picked < nmax; }
    const int lo = max(c - (spread ? rb : 0), 0), hi = c + (spread ? rf : 0);
    const bool rem = act && (FLAT || lab != 0);
    for (int t = gl; t < nwave; t += 8) {
      if (rem && t < ncand) {
        uint32_t kv, pv;
        load(t, kv, pv);
        const uint32_t loc = (FLAT ? 0xFFFFu - pv : pv) >> 6;
        if (loc - (uint32_t)lo <= (uint32_t)(hi - lo)) kill(t);
      }
    }
    if (!FLAT) {
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
    act = act && more;
  }
}

DEV_INLINE void ff_sector(const alego_params& P, int S, int E, int j, int& sp, int& ep) {
  const int NSEC = P.n_sectors;
  if (P.sector_formula == 0) { sp = (S * (NSEC - j) + E * j) / NSEC; ep = (S * (NSEC - 1 - j) + E * (j + 1)) / NSEC - 1; }   // laserOdometry.cpp:177-178
  else { const int diff = E - S; sp = S + j * diff / NSEC; ep = S + (j + 1) * diff / NSEC - 1; }                               // LO.cpp:245-249
}

// NW wavefronts: the staging / candidate part of a sector step is shared ring by ring among them, wavefront 0 picks
// BIG: sectors of more than 400 points (16 x 4000), candidate lists of up to 192 + 64 per ring sector in registers
template <int NW, bool BIG>
__global__ void __launch_bounds__(64 * NW) fe_front(DevCtx d, int sector_cap, int CS, int CF) {
  const int slot = blockIdx.y + d.slot0, ring0 = blockIdx.x * FF_G, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 3, gl = lane & 7;
  const int NS = d.NS;
  const size_t base = (size_t)slot * d.N;
  const alego_params& P = d.P;
  int* sc = d.scal + slot * SC_COUNT;
  const int M = sc[SC_M];
  const bool dbg = d.n_launch == 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char ff_smem[];
  const FfLayout L = ff_layout(sector_cap, CS, CF, NW);
  uint32_t* s_key = reinterpret_cast<uint32_t*>(ff_smem + L.off_key);     // [FF_G][ct]
  uint16_t* s_pay = reinterpret_cast<uint16_t*>(ff_smem + L.off_pay);     // [FF_G][ct]
  uint32_t* s_mark = reinterpret_cast<uint32_t*>(ff_smem + L.off_mark);   // [FF_G][mw]
  int* s_misc = reinterpret_cast<int*>(ff_smem + L.off_misc);             // [FF_G][4]: sharp candidates, flat candidates, carry
  unsigned char* wl = ff_smem + L.off_wave + wave * L.wave_bytes;
  if (blockIdx.x == 0 && threadIdx.x == 0) sc[SC_FE_EPOCH] = sc[SC_FE_EPOCH] + 1;   // (fe_ring_out of this launch tags its ring counts with it)
  // the picking wavefront's ring of this lane
  const int ring = ring0 + g;
  const bool rv = ring < NS;
  const int S = rv ? d.ring_start[slot * NS + ring] : 0, E = rv ? d.ring_end[slot * NS + ring] : 0;
  int* st = d.st_idx + ((size_t)slot * NS + (rv ? ring : 0)) * d.st_stride;
  FfPick R;
  R.act0 = false; R.sp = 0; R.carry = -1; R.n_sharp = 0; R.n_ls = 0; R.n_flat = 0;
  R.st_sharp = st; R.st_lsharp = st + d.cap_sharp; R.st_flat = st + d.cap_sharp + d.cap_lsharp;
  if (threadIdx.x < FF_G) s_misc[threadIdx.x * 4 + 2] = -1;
  __syncthreads();
  const int NSEC = P.n_sectors;
  for (int j = 0; j < NSEC; ++j) {
    // ---- candidates of sector j, ring by ring
    for (int r0 = 0; r0 < FF_G; r0 += NW) {
      const int rr = r0 + wave, rg = ring0 + rr;
      int sp = 0, ep = -1;
      if (rr < FF_G && rg < NS) ff_sector(P, d.ring_start[slot * NS + rg], d.ring_end[slot * NS + rg], j, sp, ep);
      int ns = 0, nf = 0;
      if (sp < ep) {   // (wavefront-uniform; the barriers inside ff_wide are matched by the two below)
        uint2* ovf = reinterpret_cast<uint2*>(d.st_lfds + ((size_t)slot * NS + rg) * d.H);   // the filtered-ring staging of the four-kernel path: free here
        ff_wide(d, base, M, sp, ep, s_misc[rr * 4 + 2], wl, L, s_key + rr * L.ct, s_pay + rr * L.ct, CS, CF, ovf, ovf + sector_cap, ns, nf, dbg);
      } else { __syncthreads(); __syncthreads(); }
      if (rr < FF_G && lane == 0) { s_misc[rr * 4 + 0] = ns; s_misc[rr * 4 + 1] = nf; }
      if (wave == 0) for (int w = lane; w < NW * L.mw; w += 64) s_mark[min(r0 * L.mw + w, FF_G * L.mw - 1)] = 0u;   // this step's rings' mark bitmaps
      __syncthreads();
    }
    // ---- the picks, eight rings in lock-step
    if (wave == 0) {
      int sp = 0, ep = -1;
      if (rv) ff_sector(P, S, E, j, sp, ep);
      R.act0 = rv && sp < ep; R.sp = sp;
      const int ns = R.act0 ? s_misc[g * 4 + 0] : 0, nf = R.act0 ? s_misc[g * 4 + 1] : 0;
      const uint32_t* kl = s_key + g * L.ct;
      const uint16_t* pl = s_pay + g * L.ct;
      uint32_t* mk = s_mark + g * L.mw;
      uint2* ovf = reinterpret_cast<uint2*>(d.st_lfds + ((size_t)slot * NS + (rv ? ring : 0)) * d.H);
      const int nsw = (int)wave_max_u32((uint32_t)ns), nfw = (int)wave_max_u32((uint32_t)nf);
      if (nsw > CS) ff_pick_mem<false>(P, R, ns, nsw, s_key + g * L.ct, pl, CS, ovf, mk, L.mw, gl);
      else if (nsw <= (BIG ? 64 : 32)) ff_pick_regs<(BIG ? 8 : 4), false>(P, R, ns, kl, pl, mk, L.mw, gl);
      else if (nsw <= (BIG ? 128 : 64)) ff_pick_regs<(BIG ? 16 : 8), false>(P, R, ns, kl, pl, mk, L.mw, gl);
      else if (nsw <= (BIG ? 192 : 96)) ff_pick_regs<(BIG ? 24 : 12), false>(P, R, ns, kl, pl, mk, L.mw, gl);
      else ff_pick_mem<false>(P, R, ns, nsw, s_key + g * L.ct, pl, CS, ovf, mk, L.mw, gl);
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the sharp picks' marks are in LDS before the flat candidates are tested against them
      __builtin_amdgcn_wave_barrier();
      if (nfw > CF) ff_pick_mem<true>(P, R, nf, nfw, s_key + g * L.ct + CS, pl + CS, CF, ovf + sector_cap, mk, L.mw, gl);
      else if (nfw <= 32) ff_pick_regs<4, true>(P, R, nf, kl + CS, pl + CS, mk, L.mw, gl);
      else if (BIG && nfw <= 64) ff_pick_regs<8, true>(P, R, nf, kl + CS, pl + CS, mk, L.mw, gl);
      else ff_pick_mem<true>(P, R, nf, nfw, s_key + g * L.ct + CS, pl + CS, CF, ovf + sector_cap, mk, L.mw, gl);
      if (gl == 0) s_misc[g * 4 + 2] = R.carry;
    }
    __syncthreads();
  }
  if (wave == 0 && rv && gl == 0) {
    int* c = d.st_cnt + ((size_t)slot * NS + ring) * 8;
    c[0] = R.n_sharp; c[1] = R.n_ls; c[2] = R.n_flat;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// fe_ring_out
// ---------------------------------------------------------------------------------------------------------------------------------
#define FO_BLOCK 256
#define FO_NB 256          // buckets of the run ordering (as fe_voxel)
#define FO_U 4             // loads kept in flight per thread
#define FO_CAP_PCT 72
#define FO_INVALID 0xFFFFFFFFu
#define FO_SPIN_LIMIT (1 << 22)
// points of the ring staged in LDS (by position in the ring): the rest is read from the L2 on every pass (fe_voxel's budget: the workgroup stays
// at ~40 KB so that four rings share a CU)
__host__ __device__ inline int fo_stage_cap(int H) {
  const int by_pct = (H * FO_CAP_PCT / 100 + 15) & ~15, by_lds = (40960 - 2200 - 10 * H) / 16;
  return by_lds <= 0 ? 0 : (by_pct < by_lds ? by_pct : (by_lds & ~15));
}
static size_t fo_lds_bytes(int H) { return (size_t)10 * H + (size_t)16 * fo_stage_cap(H); }

__global__ void __launch_bounds__(FO_BLOCK) fe_ring_out(DevCtx d) {
  const int slot = blockIdx.x + d.slot0, ring = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int NS = d.NS, H = d.H;
  const size_t base = (size_t)slot * d.N;
  const alego_params& P = d.P;
  int* scv = d.scal + slot * SC_COUNT;
  const int cur = cur_in_flight(d, slot);
  const size_t fb = (size_t)slot * 2 + cur;
  const unsigned epoch = (unsigned)scv[SC_FE_EPOCH] & 0xFFFFu;
  const int S = d.ring_start[slot * NS + ring], E = d.ring_end[slot * NS + ring];
  const int n_all = min(max(E - S, 0), H);               // the sectors of a ring cover [S, E - 1] (:177-178)
  const int* allc = d.st_cnt + (size_t)slot * NS * 8;
  const int* st = d.st_idx + ((size_t)slot * NS + ring) * d.st_stride;
  const float4* seg = d.seg_lo + base;
  extern __shared__ __attribute__((aligned(16))) unsigned char fo_smem[];
  float4* s_pt = reinterpret_cast<float4*>(fo_smem);                            // the ring's points [cap]
  const int cap = fo_stage_cap(H);
  unsigned char* fv2 = fo_smem + (size_t)16 * cap;
  uint32_t* s_key = reinterpret_cast<uint32_t*>(fv2);                           // voxel id per point (FO_INVALID: not in less_flat_scan)   [H]
  uint32_t* s_rvid = s_key;                                                     // voxel id per run, compacted in place (run r <= its first point)
  uint16_t* s_rstart = reinterpret_cast<uint16_t*>(fv2 + 4 * (size_t)H);        // first point of the run  [H]
  uint16_t* s_order = reinterpret_cast<uint16_t*>(fv2 + 6 * (size_t)H);         // valid runs sorted by (voxel id, run) [H]
  uint16_t* s_tmp = reinterpret_cast<uint16_t*>(fv2 + 8 * (size_t)H);           // valid runs dealt into buckets [H]
  __shared__ uint32_t s_bm[FE_MAXH / 32 + 1];     // holes of less_flat_scan: the ring's less-sharp picks (label > 0, :284) and points of skipped sectors (:181)
  __shared__ int s_wpre[FE_MAXH / 32 + 1];
  __shared__ float s_red[6][FO_BLOCK / 64];
  __shared__ int s_scan[FO_BLOCK / 64], s_cntv[FO_BLOCK / 64];
  __shared__ int s_boff[FO_NB + 1], s_bcur[FO_NB + 1];
  __shared__ int s_pre[8], s_look[3];
  // ---- offsets of the ring's sharp / less_sharp / flat picks and of its less_sharp boxes (everything fe_front counted)
  if (tid < 64) {
    const int r = tid;
    const int c0 = r < NS ? allc[r * 8 + 0] : 0, c1 = r < NS ? allc[r * 8 + 1] : 0, c2 = r < NS ? allc[r * 8 + 2] : 0;
    const int b1 = (c1 + LO_CH - 1) / LO_CH;
    int v[8] = {r < ring ? c0 : 0, r < ring ? c1 : 0, r < ring ? c2 : 0, r < ring ? b1 : 0, c0, c1, c2, b1};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
      if (lane == 0) s_pre[k] = v[k];
    }
  }
  const int BW = (n_all + 31) / 32;
  for (int w = tid; w <= BW; w += FO_BLOCK) s_bm[w] = 0u;
  __syncthreads();
  const int cnt3[3] = {allc[ring * 8 + 0], allc[ring * 8 + 1], allc[ring * 8 + 2]};
  for (int t = tid; t < cnt3[1]; t += FO_BLOCK) {
    const int i = st[d.cap_sharp + t] - S;
    if (i >= 0 && i < n_all) atomicOr(&s_bm[i >> 5], 1u << (i & 31));
  }
  if (tid < P.n_sectors) {
    int sp, ep;
    ff_sector(P, S, E, tid, sp, ep);
    const int i = sp - S;
    if (sp == ep && i >= 0 && i < n_all) atomicOr(&s_bm[i >> 5], 1u << (i & 31));   // a one-point sector is skipped (sp >= ep)
  }
  // ---- the ring's part of the sharp / less_sharp / flat clouds, its less_sharp boxes, its offsets
  {
    const int stoff[3] = {0, d.cap_sharp, d.cap_sharp + d.cap_lsharp};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float4* dst = d.feat[k] + fb * d.fcap[k] + s_pre[k];
      int* dsti = d.feat_idx[k] + fb * d.fcap[k] + s_pre[k];
      for (int t = tid; t < cnt3[k]; t += FO_BLOCK) { const int idx = st[stoff[k] + t]; dst[t] = seg[idx]; dsti[t] = idx; }
    }
    float4* bx = d.lo_box + (fb * 2 + 1) * d.lo_box_cap * 2 + (size_t)2 * s_pre[3];
    const int nb = (cnt3[1] + LO_CH - 1) / LO_CH;
    for (int b0 = 0; b0 < nb; b0 += FO_BLOCK / LO_CH) {
      const int b = b0 + tid / LO_CH, t = b * LO_CH + tid % LO_CH;
      const bool v = b < nb && t < cnt3[1];
      float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
      if (v) { const float4 p = seg[st[d.cap_sharp + t]]; mn[0] = mx[0] = p.x; mn[1] = mx[1] = p.y; mn[2] = mx[2] = p.z; }
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int o = LO_CH / 2; o > 0; o >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64)); }
      if (tid % LO_CH == 0 && v) {
        bx[2 * b] = make_float4(mn[0], mn[1], mn[2], __int_as_float(s_pre[1] + t));
        bx[2 * b + 1] = make_float4(mx[0], mx[1], mx[2], __int_as_float(min(LO_CH, cnt3[1] - t)));
      }
    }
    int* ro = d.ring_off + (fb * 2) * (NS + 1);
    int* rb = d.ring_boff + (fb * 2) * (NS + 1);
    if (tid == 0) {
      ro[ring] = s_pre[1]; rb[ring] = s_pre[3];
      if (ring == 0) {
        ro[NS] = s_pre[5]; rb[NS] = s_pre[7];
        int* fc = d.feat_cnt + fb * 4;
        fc[0] = s_pre[4]; fc[1] = s_pre[5]; fc[2] = s_pre[6];
      }
    }
  }
  __syncthreads();
  auto hole = [&](int i) -> bool { return (s_bm[i >> 5] >> (i & 31)) & 1u; };
  auto point = [&](int i) -> float4 { if (i < cap) return s_pt[i]; return seg[S + i]; };
  // ---- pcl::VoxelGrid on the ring's less_flat_scan: runs of consecutive equal voxel ids, ordered through monotone buckets,
  // centroid in original order — fe_voxel's algorithm on positions of the ring instead of an index list; a hole is a run of its own that
  // is never ordered
  const float inv = 1.0f / P.less_flat_leaf;
  float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  int nval = 0;
  for (int i0 = tid; i0 < n_all; i0 += FO_BLOCK * FO_U) {
    float4 pt[FO_U];
#pragma unroll
    for (int u = 0; u < FO_U; ++u) pt[u] = seg[S + min(i0 + u * FO_BLOCK, n_all - 1)];
#pragma unroll
    for (int u = 0; u < FO_U; ++u) {
      const int i = i0 + u * FO_BLOCK;
      if (i < min(n_all, cap)) s_pt[i] = pt[u];
      const bool v = i < n_all && !hole(i);
      nval += v ? 1 : 0;
      mn[0] = fminf(mn[0], v ? pt[u].x : 3.402823466e+38f); mn[1] = fminf(mn[1], v ? pt[u].y : 3.402823466e+38f); mn[2] = fminf(mn[2], v ? pt[u].z : 3.402823466e+38f);
      mx[0] = fmaxf(mx[0], v ? pt[u].x : -3.402823466e+38f); mx[1] = fmaxf(mx[1], v ? pt[u].y : -3.402823466e+38f); mx[2] = fmaxf(mx[2], v ? pt[u].z : -3.402823466e+38f);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nval += __shfl_xor(nval, o, 64);
  if (lane == 0) s_cntv[tid >> 6] = nval;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64)); }
    if (lane == 0) { s_red[a][tid >> 6] = mn[a]; s_red[3 + a][tid >> 6] = mx[a]; }
  }
  __syncthreads();
  nval = 0;
#pragma unroll
  for (int w = 0; w < FO_BLOCK / 64; ++w) nval += s_cntv[w];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    mn[a] = s_red[a][0]; mx[a] = s_red[3 + a][0];
#pragma unroll
    for (int w = 1; w < FO_BLOCK / 64; ++w) { mn[a] = fminf(mn[a], s_red[a][w]); mx[a] = fmaxf(mx[a], s_red[3 + a][w]); }
  }
  bool passthrough = false;
  if (nval > 0) {
    const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1, dz = (long long)((mx[2] - mn[2]) * inv) + 1;
    passthrough = dx * dy * dz > 2147483647LL;   // "leaf size too small": the input is returned unchanged
  }
  int nout = 0, nrv = 0, nruns = 0;
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
#pragma unroll 4
    for (int i = tid; i < n_all; i += FO_BLOCK) {
      const float4 q = point(i);
      const int i0 = (int)(floorf(q.x * inv) - (float)minb[0]);
      const int i1 = (int)(floorf(q.y * inv) - (float)minb[1]);
      const int i2 = (int)(floorf(q.z * inv) - (float)minb[2]);
      s_key[i] = hole(i) ? FO_INVALID : (uint32_t)(i0 + i1 * mul1 + i2 * mul2);
    }
    __syncthreads();
    // runs of consecutive equal voxel ids
    for (int c0 = 0; c0 < n_all; c0 += FO_BLOCK) {
      const int i = c0 + tid;
      const uint32_t mykey = i < n_all ? s_key[i] : 0u;   // read before the barrier: the run ids are compacted into the same array
      const bool head = i < n_all && (i == 0 || mykey != s_key[i - 1]);
      const unsigned long long m = __ballot(head);
      if (lane == 0) s_scan[tid >> 6] = (int)__popcll(m);
      __syncthreads();
      int woff = 0, tot = 0;
#pragma unroll
      for (int w = 0; w < FO_BLOCK / 64; ++w) { if (w < (tid >> 6)) woff += s_scan[w]; tot += s_scan[w]; }
      if (head) {
        const int r = nruns + woff + (int)__popcll(m & ((1ull << lane) - 1ull));
        s_rvid[r] = mykey;
        s_rstart[r] = (uint16_t)i;
      }
      nruns += tot;
      __syncthreads();
    }
    // order the valid runs by (voxel id, run index): dealt into <= FO_NB buckets monotone in the voxel id, ranked inside the bucket
    {
      unsigned T = (unsigned)divb[0] * (unsigned)divb[1] * (unsigned)divb[2];
      if (T == 0) T = 1;
      int shift = 0;
      while (((T - 1) >> shift) >= (unsigned)FO_NB) ++shift;
      const int nb = (int)((T - 1) >> shift) + 1;
      for (int b = tid; b <= nb; b += FO_BLOCK) s_boff[b] = 0;
      __syncthreads();
      for (int r = tid; r < nruns; r += FO_BLOCK) { const uint32_t v = s_rvid[r]; if (v != FO_INVALID) atomicAdd(&s_boff[min((int)(v >> shift), nb - 1) + 1], 1); }
      __syncthreads();
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
      for (int r = tid; r < nruns; r += FO_BLOCK) {
        const uint32_t v = s_rvid[r];
        if (v != FO_INVALID) s_tmp[atomicAdd(&s_bcur[min((int)(v >> shift), nb - 1)], 1)] = (uint16_t)r;  // s_bcur[b] starts at s_boff[b] (written one slot up, read one down)
      }
      __syncthreads();
      for (int t = tid; t < nrv; t += FO_BLOCK) {
        const int r = s_tmp[t];
        const uint32_t v = s_rvid[r];
        const int b = min((int)(v >> shift), nb - 1);
        const int bs = s_boff[b], be = s_boff[b + 1];
        int rank = bs;
        for (int q = bs; q < be; ++q) { const int o = s_tmp[q]; const uint32_t u = s_rvid[o]; rank += (u < v) || (u == v && o < r); }
        s_order[rank] = (uint16_t)r;
      }
    }
    __syncthreads();
    // voxels = first runs of their id in the order
    int nh = 0;
    for (int j = tid; j < nrv; j += FO_BLOCK) nh += (j == 0 || s_rvid[s_order[j]] != s_rvid[s_order[j - 1]]) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nh += __shfl_xor(nh, o, 64);
    if (lane == 0) s_cntv[tid >> 6] = nh;   // (s_cntv was last read before the barriers above)
    __syncthreads();
    nout = 0;
#pragma unroll
    for (int w = 0; w < FO_BLOCK / 64; ++w) nout += s_cntv[w];
  }
  // ---- the ring's less_flat offset: its count for the rings above, the counts of the rings below
  if (tid == 0) __hip_atomic_store(&d.fe_sync[(size_t)slot * NS + ring], (epoch << 16) | (unsigned)nout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid < 64) {
    int c = 0, nbx = 0, bad = 0;
    if (tid < ring) {
      unsigned v = 0;
      int spins = 0;
      while (true) {
        v = __hip_atomic_load(&d.fe_sync[(size_t)slot * NS + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 16) == epoch || ++spins > FO_SPIN_LIMIT) break;
        __builtin_amdgcn_s_sleep(2);
      }
      bad = (v >> 16) != epoch;
      c = (int)(v & 0xFFFFu);
      nbx = (c + LO_CH - 1) / LO_CH;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o, 64); nbx += __shfl_xor(nbx, o, 64); bad += __shfl_xor(bad, o, 64); }
    if (lane == 0) { s_look[0] = c; s_look[1] = nbx; s_look[2] = bad; }
  }
  __syncthreads();
  const int off3 = s_look[0], boff3 = s_look[1];
  if (s_look[2]) {   // never expected (a lower ring's workgroup is dispatched before this one): nothing is written, the error surfaces on the host
    if (tid == 0) scv[SC_FE_ERR] = 1;
    return;
  }
  float4* out = d.feat[F_LFLAT] + fb * d.fcap[F_LFLAT] + off3;
  if (passthrough) {
    __syncthreads();
    for (int i = tid; i < n_all; i += FO_BLOCK)
      if (!hole(i)) out[i - (s_wpre[i >> 5] + __popc(s_bm[i >> 5] & ((1u << (i & 31)) - 1u)))] = point(i);
  } else {
    // first run of every voxel -> output rank; it accumulates all runs of the voxel in order
    int nvox = 0;
    for (int c0 = 0; c0 < nrv; c0 += FO_BLOCK) {
      const int j = c0 + tid;
      const bool head = j < nrv && (j == 0 || s_rvid[s_order[j]] != s_rvid[s_order[j - 1]]);
      const unsigned long long m = __ballot(head);
      if (lane == 0) s_scan[tid >> 6] = (int)__popcll(m);
      __syncthreads();
      int woff = 0, tot = 0;
#pragma unroll
      for (int w = 0; w < FO_BLOCK / 64; ++w) { if (w < (tid >> 6)) woff += s_scan[w]; tot += s_scan[w]; }
      if (head) {
        const int rank = nvox + woff + (int)__popcll(m & ((1ull << lane) - 1ull));
        const uint32_t vid = s_rvid[s_order[j]];
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        int c = 0;
        for (int jj = j; jj < nrv && s_rvid[s_order[jj]] == vid; ++jj) {
          const int r = s_order[jj], i0 = s_rstart[r], len = (r + 1 < nruns ? (int)s_rstart[r + 1] : n_all) - i0;
          for (int i = i0; i < i0 + len; ++i) { const float4 q = point(i); sx += q.x; sy += q.y; sz += q.z; si += q.w; ++c; }   // strictly in order
        }
        const float fn = (float)c;
        out[rank] = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
      }
      nvox += tot;
      __syncthreads();
    }
  }
  __syncthreads();   // the ring's part of less_flat is written: its boxes are taken from there (global memory written by this workgroup)
  {
    float4* bx = d.lo_box + (fb * 2 + 0) * d.lo_box_cap * 2 + (size_t)2 * boff3;
    const int nb = (nout + LO_CH - 1) / LO_CH;
    for (int b0 = 0; b0 < nb; b0 += FO_BLOCK / LO_CH) {
      const int b = b0 + tid / LO_CH, t = b * LO_CH + tid % LO_CH;
      const bool v = b < nb && t < nout;
      float bn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, bxm[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
      if (v) { const float4 p = out[t]; bn[0] = bxm[0] = p.x; bn[1] = bxm[1] = p.y; bn[2] = bxm[2] = p.z; }
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int o = LO_CH / 2; o > 0; o >>= 1) { bn[a] = fminf(bn[a], __shfl_xor(bn[a], o, 64)); bxm[a] = fmaxf(bxm[a], __shfl_xor(bxm[a], o, 64)); }
      if (tid % LO_CH == 0 && v) {
        bx[2 * b] = make_float4(bn[0], bn[1], bn[2], __int_as_float(off3 + t));
        bx[2 * b + 1] = make_float4(bxm[0], bxm[1], bxm[2], __int_as_float(min(LO_CH, nout - t)));
      }
    }
    if (tid == 0) {
      int* ro = d.ring_off + (fb * 2 + 1) * (NS + 1);
      int* rb = d.ring_boff + (fb * 2 + 1) * (NS + 1);
      ro[ring] = off3; rb[ring] = boff3;
      if (ring == NS - 1) { ro[NS] = off3 + nout; rb[NS] = boff3 + nb; d.feat_cnt[fb * 4 + 3] = off3 + nout; }
      d.st_cnt[((size_t)slot * NS + ring) * 8 + 4] = nout;
    }
  }
  // ---- cloud_label_ of the ring's points for the single-scan entry points / tests (:196-204,:245): 2 sharp, 1 less sharp, -1 flat
  if (d.n_launch == 1) {
    const int rf = S - 5, cntr = E - S + 11;
    for (int k = tid; k < cntr; k += FO_BLOCK) if (rf + k >= 0 && rf + k < d.N) d.plabel[base + rf + k] = 0;
    __syncthreads();
    for (int t = tid; t < cnt3[1]; t += FO_BLOCK) d.plabel[base + st[d.cap_sharp + t]] = 1;
    for (int t = tid; t < cnt3[2]; t += FO_BLOCK) d.plabel[base + st[d.cap_sharp + d.cap_lsharp + t]] = -1;
    __syncthreads();
    for (int t = tid; t < cnt3[0]; t += FO_BLOCK) d.plabel[base + st[t]] = 2;
  }
}

void launch_fe_curv_debug(const DevCtx& d, hipStream_t st);   // kernels_fe.hip: fe_curv alone (curvature sums / occlusion marks of the points outside every sector, tests only)

// fused path: everything but alego_params.sort_mode = 2 (the libstdc++ tie order needs the whole sector's keys: four-kernel path)
bool fe_fused_eligible(const DevCtx& d) { return d.opt_fe_fused && !d.opt_fe_pick1 && d.P.sort_mode != 2 && d.NS <= 64 && d.H <= FE_MAXH; }

void launch_fe_fused(const DevCtx& d, hipStream_t st) {
  static const bool cfg = hipFuncSetAttribute(reinterpret_cast<const void*>(fe_ring_out), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fo_lds_bytes(FE_MAXH)) == hipSuccess;
  (void)cfg;
  const int sector_cap = (d.H + d.P.n_sectors - 1) / (d.P.n_sectors > 0 ? d.P.n_sectors : 1) + 2;
  // candidates of a ring sector kept in LDS: 96 sharp + 32 flat cover every sector of the 16 x 1800 and 64 x 2048 streams (sharp: mean 32, max 93);
  // a 16 x 4000 sector has up to ~140 sharp and ~200 flat candidates.  More than that spills to HBM and is picked from there.
  int CS = sector_cap > 400 ? 192 : 96, CF = sector_cap > 400 ? 64 : 32;
  if (d.opt_fe_cand > 0) { CS = std::min(d.opt_fe_cand, CS); CF = std::min((d.opt_fe_cand + 1) / 2, CF); }   // (tests: small lists drive the overflow path)
  CS = (CS + 1) & ~1; CF = (CF + 1) & ~1;
  if (d.n_launch == 1) launch_fe_curv_debug(d, st);
  const FfLayout L = ff_layout(sector_cap, CS, CF, 1);
  const dim3 gf((d.NS + FF_G - 1) / FF_G, d.n_launch);
  if (sector_cap > 400) { ALEGO_LAUNCH((fe_front<1, true>), gf, dim3(64), (size_t)L.total, st, d, sector_cap, CS, CF); }
  else { ALEGO_LAUNCH((fe_front<1, false>), gf, dim3(64), (size_t)L.total, st, d, sector_cap, CS, CF); }
  ALEGO_LAUNCH(fe_ring_out, dim3(d.n_launch, d.NS), dim3(FO_BLOCK), fo_lds_bytes(d.H), st, d);
}
