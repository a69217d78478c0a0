// mckpp_kernels_ps.hip - the column step: packed, stateless-lane cooperative kernel.
//
// One launch takes every resident column through mckpp_physics_ocnstep (+ check_profile), or through
// mckpp_initialize_ocean_model's per-column part, or one vmix(+ocnint) pass (p.mode).  A persistent grid of
// workgroups, each with W column slots fed from an atomic queue; a column's iteration is a sequence of
// level-parallel phases (L1..L7: every (slot, level) item of the workgroup, strided over its threads) and
// serial phases (M0..M5: the manager wave, one lane per slot or per (slot, tridiagonal system)), separated by
// workgroup barriers.
//
// A level thread keeps (almost) NOTHING between phases: the iterate (U,V,T,S of the under-relaxation) lives
// in a small scratch block per (workgroup, slot) in global memory - but for a thread's first item, whose four
// values stay in registers - and every other phase-crossing value in one of NINE rows of the slot's LDS block
// (each row is reused three to five times in a pass, see the table below).  So a workgroup serves as many
// slots as LDS holds, whatever its number of waves, and the manager wave's serial phases - the latency floor of
// a pass - cost the same for 15 slots as for 1.  While the manager wave runs the last of them (the V sweep)
// the other waves already run the next pass's L1 (all of it but the relaxation of V) for the slots that go on.
// (Round-2 history, measured on 1e5 columns x 60 levels: one wavefront per column with 13 LDS rows 1.9e7
// column-steps/s; nzp1+2 lanes per column for the whole step 1.8e7; this kernel 2.7e7 - DESIGN.md section 1.)
#include "mckpp_colmath.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {

using namespace mckpp_dev;

enum { PS_EMPTY = 0, PS_ACTIVE = 1, PS_DONE = 2, PS_WAIT = 3 /* holds a ticket whose column's previous step is still running elsewhere */ };
enum { F_NONE = 0, F_TRAP = 1, F_FINAL = 2 };

// per-slot double record
enum {
  C_B0 = 0, C_B0SOL, C_USTAR, C_UFRAC, C_UCUBE, C_HEK, C_WU01, C_WU02, C_WX01, C_WX02, C_WXNT0, C_UREFNZ, C_VREFNZ,
  C_RHO0CP0, C_RRC, X_T1 /* level-1 T and S the last L1 worked on */, X_S1, X_RHOH2O, X_RHOB,
  C_F, C_HMIXE, C_HMIXN, C_HBL, C_RHBL, C_STABLE, C_BFSFC, C_CASEA,
  C_GAT1, C_DAT1 = C_GAT1 + 3, C_DKM1 = C_DAT1 + 3,
  C_SREF = C_DKM1 + 3, C_SSURF, C_OCDEPTH, C_SFLUX1, C_SFLUX2, C_SFLUX3, C_SFLUX4, C_SFLUX5, C_SFLUX6,
  C_T1X /* + parity: the level-1 temperature of the iterate, for the two EOS items */,
  C_JRFAC = C_T1X + 2, C_JA1, C_JA2, C_JRA1, C_JRA2 /* the column's Jerlov constants (swfrac_mod.F90:59-61), fetched once per column */,
  C_COUNT_USED,
  // the manager's lanes read one field of fifteen records at once: an ODD record length spreads them over the banks
  // (52 doubles put eight lanes on every bank: 8x the bank-conflict cycles of round 3's 47, profiles/r04)
  C_COUNT = C_COUNT_USED | 1
};
// per-slot int record
enum {
  I_STATE = 0, I_ACT, I_COL, I_OLD, I_NEW, I_JER, I_INITFLAG, I_STATUS, I_NPASS, I_NPASS_TRY, I_ICONV, I_COMP, I_KMIXN,
  I_KBL, I_NRESET, I_FIN, I_MAYBE, I_LOAD /* 1: new column, 2: restart the iteration (trap retry) */, I_JU,
  I_KBLC, I_NVIOL, I_NOVER, I_NU, I_NV, I_NF, I_BAD, I_L1A /* L1 but for V done ahead, during the V sweep */,
  I_MAYBE_NEXT, I_LOCEAN, I_PAR /* which C_T1X holds the iterate's level-1 temperature */,
  I_TINY /* some whole-layer term of the reference-level sums is a tiny non-zero number (L2) */,
  I_STEP /* which step of the launch this column is in (0 .. nsteps_launch-1) */,
  I_STRAG /* counted as a straggler: past its solo_after-th pass of a try, or at itermax in its previous step (M0) */, I_COUNT_USED,
  I_COUNT = I_COUNT_USED | 1   // odd, like C_COUNT: 32 ints would put every lane's record on the same two banks
};
// LDS rows of a slot and what each holds between which phases of a pass:
//   Q_DM   (LDD talpha L1..L2; else the whole-layer terms of the reference-level sum of U, L2)  difm: interior L3;
//          from L5 p = tri(:,1) difm of the momentum system (to L7)
//   Q_DT   Ritop L2..L3; dift interior L3; from L5 p of the T system (without double diffusion: of T and S, whose
//          difs = dift bit for bit, one factorisation for both) to M4; refined reciprocals of the momentum
//          pivots L7..M5 (V sweep)
//   Q_DS   dVsq L2..L3; with double diffusion difs L3, p of the S system L5..M4; without it dift for L6 (L5..L6),
//          then gam of the momentum system M4..M5
//   Q_YU   previous U solution .. L1; U of the iterate L1..L2; Monin-Obukhov depth L3..L4; rhs L6; solution M4
//   Q_YT   previous T solution .. L1; dbloc L2..L3; rhs L6; solution M4
//   Q_YS   previous S solution .. L1; buoyancy L1..L2; hmin candidates L4..M3; rhs L6; solution M4
//   Q_YV   previous V solution .. L1; V of the iterate L1..L2; bulk Ri L3..L4 (scan M2); ghat L5..L6;
//          rhs L7; solution M5
//   Q_GM   Rig L2..L3; q = tri(:,0) difm(k-1) of the momentum system L5..M5 (U and V sweeps)
//   Q_BET  (LDD T L1..L2; else the whole-layer terms of V, L2) q of the T (and S) system L5..M4, which its gam
//          overwrites level by level; pivots of the momentum system L7..M5 (V sweep)
// In the instability trap Q_DM, Q_DT, Q_DS, Q_GM carry the four rmsd terms, in the isotherm check Q_DM, Q_DT.
// Optional-physics builds: rho, cp (L1..L6); with double diffusion also Q_X1, Q_X2: alphaDT, betaDS (L2..L3), then
// dift, difs for L6; Q_S1, Q_S2: LDD sbeta and S (L1..L2), then q / gam of the S system and gam of the momentum system.  Three kernel variants XV: 0 default physics (9 rows), 1 optional physics (11), 2 optional physics
// with double diffusion (15).
enum { Q_DM = 0, Q_DT, Q_DS, Q_YU, Q_YT, Q_YS, Q_YV, Q_GM, Q_BET, Q_COUNT,
       Q_RHO = Q_COUNT, Q_CP, Q_COUNT_EXT, Q_X1 = Q_COUNT_EXT, Q_X2, Q_S1, Q_S2, Q_COUNT_EXT_DD };
// LDS layout.  A slot's rows are interleaved per level: element (row a, level i) sits at i*ROWS + a
// doubles, so a level lane reaches all its rows and the rows of its neighbours through ONE base register
// plus immediate offsets (the column depth, hence any row-major row length, is a run-time value).  ROWS is
// odd (9 / 11 / 15): 32 consecutive levels fall on 32 distinct banks.  The grid constants are interleaved the
// same way with a stride of 7.  Host and device agree on the sizes through these:
enum { K_ZM = 0, K_HM, K_T0, K_T1, K_RDZ, K_DTOHK, K_STRIDE = 7 };
__host__ __device__ inline int ps_rows(int xv) { return xv == 2 ? (int)Q_COUNT_EXT_DD : xv == 1 ? (int)Q_COUNT_EXT : (int)Q_COUNT; }
__host__ __device__ inline int ps_nl(int L) { return L; }   // level indices 0..nzp1+1 (L = nzp1+2 items per column)
// Slot stride (doubles) of a workgroup of W slots.  LDS has 64 banks of 4 bytes; an 8-byte access of a wave goes
// through in two halves of 32 lanes, each conflict-free if its lanes' double-word addresses differ mod 32.  The
// level-major phases (L2..L5) take the items in the order (level, slot) - consecutive lanes are consecutive slots
// of one level, then the next level - so lane j of a half-wave sits at slot*SS + level*ROWS: an arithmetic
// progression mod 32 exactly when W*SS = ROWS (mod 32), and then - SS odd - a permutation of the 32 residues: no
// conflict at all.  (Round 3 had SS = ROWS*L rounded up to odd: at 69 levels SS = ROWS = 9 mod 32, every lane of a
// level conflicting with its neighbour slot of the next - 0.59 of the LDS-active cycles were bank conflicts, 0.22 at
// 60 levels.)  Of the 32 paddings the one with the fewest conflicts of that pattern is taken, then of the manager
// lanes' pattern (slot*SS + row of the system), then the smallest; at most 31 doubles per slot.
// MEASURED (r04, profiles/r04/experiments/slot_stride.txt): the conflict-free stride changes the rate by -1.7 ... +0.5 %
// (60, 69, 100 levels, both solver modes; the shipped-namelist shape within noise): bank conflicts are not what
// these phases wait for.  So the default stays round 3's rule - rows*L rounded up to odd, not +-1 mod 32 (the manager
// lanes' pattern) - and MCKPP_PS_CONFLICT_FREE=1 selects the stride below (the counters of both are on record).
__host__ __device__ constexpr int ps_ss_default(int s0)   // the default rule: rows * L rounded up to odd, not +-1 mod 32
{
  int s = s0;
  if (!(s & 1)) ++s;
  while ((s & 31) == 1 || (s & 31) == 31) s += 2;
  return s;
}
__host__ inline int ps_ss(int L, int xv, int W)
{
  const int rows = ps_rows(xv), s0 = rows * ps_nl(L);
  static const bool conflict_free = getenv("MCKPP_PS_CONFLICT_FREE") != nullptr && atoi(getenv("MCKPP_PS_CONFLICT_FREE")) != 0;
  if (!conflict_free) return ps_ss_default(s0);
  if (W <= 1) return s0 | 1;
  int best = s0, best_cost = 1 << 30;
  for (int pad = 0; pad < 32; ++pad) {
    const int ss = s0 + pad;
    int lm = 0;   // level-major: the worst multiplicity of a residue in a half-wave, summed over its alignments
    for (int start = 0; start < W * 8; start += 8) {
      int cnt[32] = {0}, worst = 0;
      for (int j = start; j < start + 32; ++j) {
        const int b = ((j % W) * ss + (j / W) * rows) & 31;
        if (++cnt[b] > worst) worst = cnt[b];
      }
      lm += worst;
    }
    int mg = 0;   // manager lanes (slot, system) of the sweeps: solution rows Q_YU + system
    for (int half = 0; half < 2; ++half) {
      int cnt[32] = {0}, worst = 0;
      for (int l = 32 * half; l < 32 * half + 32 && l < 3 * W; ++l) {
        const int b = ((l / 3) * ss + (int)Q_YU + l % 3) & 31;
        if (++cnt[b] > worst) worst = cnt[b];
      }
      mg += worst;
    }
    const int cost = lm * 64 + mg * 8 * W + pad;   // (lm is a sum over W alignments)
    if (cost < best_cost) { best_cost = cost; best = ss; }
  }
  return best;
}
__host__ __device__ inline int ps_scratch_ld(int nzp1) { return (nzp1 + 7) & ~7; }   // row length of the iterate's scratch
__host__ inline size_t ps_lds_bytes(int L, int W, int xv)
{
  return (size_t)(K_STRIDE * ps_nl(L) + 2 + W * ps_ss(L, xv, W) + W * C_COUNT) * sizeof(double) +
         (size_t)(W * I_COUNT + 32) * sizeof(int);
}


// ---- the manager wave's serial sweeps, one lane per slot (x system) over the slots' level-interleaved rows
// (element (row a, level i) of slot s at slots[s*SS + i*KS + a]; grid constants with stride CS).
// What a sweep costs (tools/ubench/sweeps.hip, lds.hip: one wave, any number of active lanes): 6 cycles per fp64
// instruction whether or not it depends on the one before, 25 for v_rcp_f64, ~8 per LDS read instruction, ~17 per
// 8-byte LDS store (34 per ds_write2_b64), 35-50 for a scalar branch on a fresh vector compare - plus the full LDS
// latency (64-100 cycles) wherever a read is waited for right after it was issued.  So: everything that is not on
// a recurrence is done elsewhere (by all threads, in a level phase), operands are fetched a trip ahead, a level
// stores as little as it can, and the conditions that need other arithmetic are tested once per trip of two
// levels, after the fact (the trip is then redone from the state it started with).  Running the factorisation
// and the forward solution as two instruction streams on two waves (the solver following the factoriser through
// a progress word in LDS) was measured and dropped: each stream alone is 220 / 260 cycles per level against 280
// fused, the pair 305 - both wait on the same LDS stores.

// ---- LDS accesses the compiler does not schedule.  The short recurrences (bulk-Ri scan, back substitution) are
// bound by the latency of their LDS reads (a trip of four levels is 50 cycles of arithmetic): the operands of the
// NEXT trip must be in flight while the current one computes.  Written in C++ the compiler's wait-count insertion
// at the loop header waits for them together with the current ones (measured: slower than no prefetch).  So these
// loops issue their reads through the helpers below and say themselves when a value is needed: ps_lds_read2
// starts a read of two doubles at (addr + 8*OFF0, addr + 8*OFF1) bytes; ps_lds_wait<N>(values...) returns once
// at most N LDS instructions issued after the ones that fetch `values` are still outstanding (LDS returns in
// order).  The values are tied through the wait statement, so nothing can use them before it.
typedef double ps_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned ps_lds_addr(const double *p) { return (unsigned)(unsigned long long)p; }   // LDS offset = low half of its flat address
template <int OFF0, int OFF1>
__device__ __forceinline__ ps_d2 ps_lds_read2(unsigned addr)
{
  ps_d2 v;
  asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(OFF0), "n"(OFF1) : "memory");
  return v;
}
template <int N> __device__ __forceinline__ void ps_lds_wait(ps_d2 &a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N> __device__ __forceinline__ void ps_lds_wait(ps_d2 &a, ps_d2 &b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
template <int N> __device__ __forceinline__ void ps_lds_wait(ps_d2 &a, ps_d2 &b, ps_d2 &c)
{
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N));
}
template <int N> __device__ __forceinline__ void ps_lds_wait(ps_d2 &a, ps_d2 &b, ps_d2 &c, ps_d2 &d)
{
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
// The stores of such a loop go the same way (a compiler-issued LDS store in the loop makes the compiler wait for
// everything outstanding at the loop header); ps_lds_drain() after the loop: every LDS access issued here has
// retired (the compiler does not know of them, so nothing else would wait before the next barrier).
template <int OFF0, int OFF1>
__device__ __forceinline__ void ps_lds_write2(unsigned addr, double a, double b)
{
  asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" : : "v"(addr), "v"(a), "v"(b), "n"(OFF0), "n"(OFF1) : "memory");
}
__device__ __forceinline__ void ps_lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// MAX(a, b) as one v_max_f64 where b is never NaN and never -0 (the scan below: b = rb + 1e-16 with rb a previous
// result, never NaN because a NaN `a` is passed over either way): then it returns what a > b ? a : b returns -
// compare, two selects - for every a, at a third of the dependent latency.
__device__ __forceinline__ double ps_max(double a, double b)
{
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// bldepth_mod.F90:137: Rib(ku) = MAX(Rib(ku), Rib(ka) + epsln) down the column, four levels per trip, the next
// trip's values fetched before the current trip's recurrence runs
// Only the levels down to the first one whose running maximum exceeds Ricr matter: L4 looks for the shallowest
// level that satisfies one of its criteria, the interpolated depth of that crossing (bldepth_mod.F90:139-150) is
// one of them and lies within the level where it happens or the next (a quotient that rounds to one), so whatever
// the rows hold further down cannot win.  The scan stops eight levels after the last of the workgroup's columns
// has crossed (the running maximum never falls); a column that never crosses keeps it going to the bottom.
// The levels k0 .. nz of the rows (k0 = 2, or where an earlier call ended, whose last value is the running maximum
// to go on from); L3 has formed the bulk Richardson numbers down to level nz only.  out[0]: the deepest level whose
// running maximum is in place; out[1]: 1 if the scan stopped because every column had crossed.
__device__ __forceinline__ void ps_scan_rib(int W, int row, double *slots, int SS, int KS, int k0, int nz, const int *sact,
                                            int sact_stride, int lane, double Ricr, int *out)
{
  const double epsln16 = 1.e-16;
  asm volatile("" : "+v"(lane));
  if (lane < W && sact[lane * sact_stride]) {
    double *r = slots + lane * SS + row;
    double rb = k0 > 2 ? r[(k0 - 1) * KS] : 0.0;
    asm volatile("" : "+v"(rb));   // its wait here, not at its first use inside the loop
    int k = k0;
    bool stop = false;   // every column of the wave had crossed before the last eight levels scanned
    if (k + 3 <= nz) {   // a trip: levels k .. k+3; KS = 9, 11 or 15 doubles per level
      const unsigned step = 4u * (unsigned)KS * 8u;
      unsigned ad = ps_lds_addr(r + k * KS);
      auto rd_lo = [&](unsigned a) { return KS == 9 ? ps_lds_read2<0, 9>(a) : KS == 11 ? ps_lds_read2<0, 11>(a) : ps_lds_read2<0, 15>(a); };
      auto rd_hi = [&](unsigned a) { return KS == 9 ? ps_lds_read2<18, 27>(a) : KS == 11 ? ps_lds_read2<22, 33>(a) : ps_lds_read2<30, 45>(a); };
      auto body = [&](int kk, const ps_d2 &v01, const ps_d2 &v23) {
        rb = ps_max(v01.x, rb + epsln16); const double b0 = rb;
        rb = ps_max(v01.y, rb + epsln16); const double b1 = rb;
        rb = ps_max(v23.x, rb + epsln16); const double b2 = rb;
        rb = ps_max(v23.y, rb + epsln16);
        const unsigned aw = ps_lds_addr(r + kk * KS);
        if (KS == 9) { ps_lds_write2<0, 9>(aw, b0, b1); ps_lds_write2<18, 27>(aw, b2, rb); }
        else if (KS == 11) { ps_lds_write2<0, 11>(aw, b0, b1); ps_lds_write2<22, 33>(aw, b2, rb); }
        else { ps_lds_write2<0, 15>(aw, b0, b1); ps_lds_write2<30, 45>(aw, b2, rb); }
      };
      ps_d2 a01 = rd_lo(ad), a23 = rd_hi(ad), b01, b23;
      while (k + 11 <= nz && !stop) {   // this trip, the next, and one more after it
        const double rb_in = rb;
        b01 = rd_lo(ad + step); b23 = rd_hi(ad + step);
        ps_lds_wait<2>(a01, a23);
        body(k, a01, a23);
        a01 = rd_lo(ad + 2 * step); a23 = rd_hi(ad + 2 * step);
        ps_lds_wait<2>(b01, b23);
        body(k + 4, b01, b23);
        k += 8; ad += 2 * step;
        stop = __builtin_amdgcn_ballot_w64(!(rb_in > Ricr)) == 0ull;
      }
      // (past the loop nothing stays in flight across a branch: see ps_backsub)
      ps_lds_wait<0>(a01, a23);   // on a stop: fetched, not needed
      if (!stop) {
        body(k, a01, a23);
        k += 4;
        if (k + 3 <= nz) {
          b01 = rd_lo(ad + step); b23 = rd_hi(ad + step);
          ps_lds_wait<0>(b01, b23);
          body(k, b01, b23);
          k += 4;
        }
      }
      ps_lds_drain();
    }
    if (!stop) {
#pragma nounroll
      for (; k <= nz; ++k) {   // the last one to three levels
        rb = dmax2(r[k * KS], rb + epsln16);
        r[k * KS] = rb;
      }
    }
    out[0] = stop ? k - 1 : nz; out[1] = stop ? 1 : 0;   // the same for every column of the wave
  }
}

// back substitution yn(i) = yn(i) - gam(i+1) yn(i+1) (solvers.F90:156-158), four levels per trip, the next trip's
// operands fetched before the current trip's recurrence runs
__device__ __forceinline__ void ps_backsub_from(double *y, const double *gm, int KS, int nz, double yy)
{
  int i = nz - 1;
  asm volatile("" : "+v"(yy));   // its wait here, not at the first use inside the loop (where it would wait for the loop's own reads too)
  if (i >= 4) {   // a trip: levels i, i-1, i-2, i-3; KS = 9, 11 or 15 doubles per level
    const unsigned step = 4u * (unsigned)KS * 8u;
    unsigned ay = ps_lds_addr(y + (i - 3) * KS), ag = ps_lds_addr(gm + (i - 2) * KS);
    auto rd_lo = [&](unsigned a) { return KS == 9 ? ps_lds_read2<0, 9>(a) : KS == 11 ? ps_lds_read2<0, 11>(a) : ps_lds_read2<0, 15>(a); };
    auto rd_hi = [&](unsigned a) { return KS == 9 ? ps_lds_read2<18, 27>(a) : KS == 11 ? ps_lds_read2<22, 33>(a) : ps_lds_read2<30, 45>(a); };
    // y10 = (y(i-1), y(i)), y32 = (y(i-3), y(i-2)); g10 = (gam(i), gam(i+1)), g32 = (gam(i-2), gam(i-1))
    auto body = [&](int ii, const ps_d2 &y32, const ps_d2 &y10, const ps_d2 &g32, const ps_d2 &g10) {
      yy = y10.y - g10.y * yy; const double r0 = yy;
      yy = y10.x - g10.x * yy; const double r1 = yy;
      yy = y32.y - g32.y * yy; const double r2 = yy;
      yy = y32.x - g32.x * yy;
      const unsigned aw = ps_lds_addr(y + (ii - 3) * KS);
      if (KS == 9) { ps_lds_write2<18, 27>(aw, r1, r0); ps_lds_write2<0, 9>(aw, yy, r2); }
      else if (KS == 11) { ps_lds_write2<22, 33>(aw, r1, r0); ps_lds_write2<0, 11>(aw, yy, r2); }
      else { ps_lds_write2<30, 45>(aw, r1, r0); ps_lds_write2<0, 15>(aw, yy, r2); }
    };
    ps_d2 a0 = rd_lo(ay), a1 = rd_hi(ay), a2 = rd_lo(ag), a3 = rd_hi(ag), b0, b1, b2, b3;
    while (i >= 12) {   // this trip, the next, and one more after it
      b0 = rd_lo(ay - step); b1 = rd_hi(ay - step); b2 = rd_lo(ag - step); b3 = rd_hi(ag - step);
      ps_lds_wait<4>(a0, a1, a2, a3);
      body(i, a0, a1, a2, a3);
      a0 = rd_lo(ay - 2 * step); a1 = rd_hi(ay - 2 * step); a2 = rd_lo(ag - 2 * step); a3 = rd_hi(ag - 2 * step);
      ps_lds_wait<4>(b0, b1, b2, b3);
      body(i - 4, b0, b1, b2, b3);
      i -= 8; ay -= 2 * step; ag -= 2 * step;
    }
    // Past the loop nothing stays in flight across a branch: between a read and its wait the value is an ordinary
    // variable for the compiler, and a copy it makes there (it does where paths merge) copies what the register
    // held before the read lands - the hardware does not interlock LDS returns (tools/check_inflight.py looks for
    // such copies in the generated code).
    ps_lds_wait<0>(a0, a1, a2, a3);
    body(i, a0, a1, a2, a3);
    i -= 4;
    if (i >= 4) {
      b0 = rd_lo(ay - step); b1 = rd_hi(ay - step); b2 = rd_lo(ag - step); b3 = rd_hi(ag - step);
      ps_lds_wait<0>(b0, b1, b2, b3);
      body(i, b0, b1, b2, b3);
      i -= 4;
    }
    ps_lds_drain();
  }
#pragma nounroll
  for (; i >= 1; --i) {   // the last one to three levels
    yy = y[(i) * KS] - gm[(i + 1) * KS] * yy;
    y[(i) * KS] = yy;
  }
}
__device__ __forceinline__ void ps_backsub(double *y, const double *gm, int KS, int nz) { ps_backsub_from(y, gm, KS, nz, y[(nz) * KS]); }

// The same away from the middle of a column, downwards: y(i) = y(i) - g(i) y(i-1), i = k0 .. nz, from yy = y(k0-1)
// (solver mode 1: the lower half of the two-ended elimination, below)
__device__ __forceinline__ void ps_fwdsub_from(double *y, const double *gm, int KS, int k0, int nz, double yy)
{
  int i = k0;
  asm volatile("" : "+v"(yy));
  if (i + 3 <= nz) {   // a trip: levels i .. i+3
    const unsigned step = 4u * (unsigned)KS * 8u;
    unsigned ay = ps_lds_addr(y + i * KS), ag = ps_lds_addr(gm + i * KS);
    auto rd_lo = [&](unsigned a) { return KS == 9 ? ps_lds_read2<0, 9>(a) : KS == 11 ? ps_lds_read2<0, 11>(a) : ps_lds_read2<0, 15>(a); };
    auto rd_hi = [&](unsigned a) { return KS == 9 ? ps_lds_read2<18, 27>(a) : KS == 11 ? ps_lds_read2<22, 33>(a) : ps_lds_read2<30, 45>(a); };
    // y01 = (y(i), y(i+1)), y23 = (y(i+2), y(i+3)); g01, g23 likewise
    auto body = [&](unsigned aw, const ps_d2 &y01, const ps_d2 &y23, const ps_d2 &g01, const ps_d2 &g23) {
      yy = y01.x - g01.x * yy; const double r0 = yy;
      yy = y01.y - g01.y * yy; const double r1 = yy;
      yy = y23.x - g23.x * yy; const double r2 = yy;
      yy = y23.y - g23.y * yy;
      if (KS == 9) { ps_lds_write2<0, 9>(aw, r0, r1); ps_lds_write2<18, 27>(aw, r2, yy); }
      else if (KS == 11) { ps_lds_write2<0, 11>(aw, r0, r1); ps_lds_write2<22, 33>(aw, r2, yy); }
      else { ps_lds_write2<0, 15>(aw, r0, r1); ps_lds_write2<30, 45>(aw, r2, yy); }
    };
    ps_d2 a0 = rd_lo(ay), a1 = rd_hi(ay), a2 = rd_lo(ag), a3 = rd_hi(ag), b0, b1, b2, b3;
    while (i + 11 <= nz) {   // this trip, the next, and one more after it
      b0 = rd_lo(ay + step); b1 = rd_hi(ay + step); b2 = rd_lo(ag + step); b3 = rd_hi(ag + step);
      ps_lds_wait<4>(a0, a1, a2, a3);
      body(ay, a0, a1, a2, a3);
      a0 = rd_lo(ay + 2 * step); a1 = rd_hi(ay + 2 * step); a2 = rd_lo(ag + 2 * step); a3 = rd_hi(ag + 2 * step);
      ps_lds_wait<4>(b0, b1, b2, b3);
      body(ay + step, b0, b1, b2, b3);
      i += 8; ay += 2 * step; ag += 2 * step;
    }
    // (past the loop nothing stays in flight across a branch: see ps_backsub_from)
    ps_lds_wait<0>(a0, a1, a2, a3);
    body(ay, a0, a1, a2, a3);
    i += 4;
    if (i + 3 <= nz) {
      b0 = rd_lo(ay + step); b1 = rd_hi(ay + step); b2 = rd_lo(ag + step); b3 = rd_hi(ag + step);
      ps_lds_wait<0>(b0, b1, b2, b3);
      body(ay + step, b0, b1, b2, b3);
      i += 4;
    }
    ps_lds_drain();
  }
#pragma nounroll
  for (; i <= nz; ++i) {   // the last one to three levels
    yy = y[(i) * KS] - gm[(i) * KS] * yy;
    y[(i) * KS] = yy;
  }
}

// The rows of tridiagonal system `sys` (0 momentum, 1 temperature, 2 salinity) of kernel variant XV.  L5, which forms
// the final diffusivities, leaves the sweeps their products with the grid's tri(:,0:1): p(i) = tri(i,1) diff(i)
// and q(i) = tri(i,0) diff(i-1) (= -cl(i), -cu(i) of tridcof, solvers.F90:28-40; the level that owns diff(i)
// writes p(i) and q(i+1)), so a level of the sweep fetches two operands less and multiplies twice less.  gam goes
// over q - except the momentum system's, whose q the V sweep needs again: its gam has a row of its own.  The
// pivots are not stored (a store costs the sweep as much as three fp64 operations): what the V sweep needs of
// them, L7 forms again from p, q and gam - bet(1) = 1 + p(1), bet(i) = ((1 + p(i)) + q(i)) + q(i) gam(i), the
// sweep's own operations on the sweep's own operands.  Without double diffusion dift and difs are the same numbers
// (rimix_mod.F90:95-97 sets them equal, blmix and enhance treat them alike), so T and S share p, q and one
// factorisation: both lanes form the same gam and pivots and store them to the same places.  With double
// diffusion S has its own rows among the staging rows that are dead by then.  L6 reads the diffusivities of T
// and S themselves (rows dl6).
template <int XV> struct ps_sysrows {
  static constexpr bool DD = XV == 2;
  __device__ static __forceinline__ int p(int sys) { return sys == 0 ? (int)Q_DM : (DD && sys == 2) ? (int)Q_DS : (int)Q_DT; }
  __device__ static __forceinline__ int q(int sys) { return sys == 0 ? (int)Q_GM : (DD && sys == 2) ? (int)Q_S1 : (int)Q_BET; }
  __device__ static __forceinline__ int gam(int sys) { return sys == 0 ? gam_m : q(sys); }
  static constexpr int gam_m = DD ? (int)Q_S2 : (int)Q_DS;   // gam of the momentum system, M4..M5
  static constexpr int dl6_t = DD ? (int)Q_X1 : (int)Q_DS, dl6_s = DD ? (int)Q_X2 : (int)Q_DS;   // dift, difs L5..L6
};

// tridcof + tridmat, forward part (solvers.F90:14-44, 112-154) for U, T, S, skewed by one level: iteration i forms
// gam(i) = cl(i-1)/bet(i-1) and y(i-1) = num(i-1)/bet(i-1) over the same denominator.  lane = (slot, system).
template <int XV>
__device__ __forceinline__ void ps_thomas_uts_fwd(int W, double *slots, int SS, int nz, const int *sact, int sact_stride,
                                                  int *sbad, int sbad_stride, int lane)
{
  constexpr int KS = XV == 2 ? (int)Q_COUNT_EXT_DD : XV == 1 ? (int)Q_COUNT_EXT : (int)Q_COUNT;
  asm volatile("" : "+v"(lane));
  if (lane < 3 * W) {
    const int sl = lane / 3, sys = lane - 3 * sl;
    if (sact[sl * sact_stride]) {
      double *base = slots + sl * SS;
      // (without double diffusion q sits seven rows after p in either system: one base, so that the two come with
      // one ds_read2_b64)
      static_assert(XV == 2 || ((int)Q_GM - (int)Q_DM == 7 && (int)Q_BET - (int)Q_DT == 7), "p and q rows");
      const double *pb = base + ps_sysrows<XV>::p(sys), *qq = XV == 2 ? base + ps_sysrows<XV>::q(sys) : pb + 7;
      double *y = base + (Q_YU + sys), *gm = base + ps_sysrows<XV>::gam(sys);
      int bad = 0;
      // The coefficients of tridcof share their products: with p(i) = tri(i,1) diff(i) and q(i) = tri(i,0) diff(i-1)
      //   cl(i) = -p(i), cu(i) = -q(i), cc(i) = (1 + p(i)) + q(i)      (solvers.F90:28-40, same roundings: a
      //   negation is exact); cu*x is -(q*x) exactly, hence cc - cu*gam = cc + q*gam.
      double pm1 = pb[(1) * KS];           // p(1)
      double bet = 1. + pm1;               // cc(1)
      double ynum = y[(1) * KS];           // y(1) = rhs(1)/bet, formed in the next level's step
      // One level: the pivot chain bet -> 1/bet -> gam -> bet' interleaved with the solution chain, both on
      // div_fast.  Two conditions need other arithmetic - a zero pivot (solvers.F90:140-151 would stop there), or a
      // tiny non-zero solution numerator that div_fast must not see; they practically never occur, so a trip of two
      // levels runs straight through, the state each of its levels started from is tested once at its end, and if
      // any lane of the wave was in either condition the trip is redone from its saved state with IEEE sequences.
      auto level = [&](int i, double p, double q, double rhs, auto slow) {
        if (slow.value && bet == 0.) { bad = 1; bet = 1.E-12; }
        const double clm1 = -pm1;
        const double cc = (1. + p) + q;
        const double rb = rcp_refine(bet);
        const double g = slow.value ? div_by_refined(clm1, bet, rb) : div_fast(clm1, bet, rb);
        const double yprev = slow.value ? div_by_refined(ynum, bet, rb) : div_fast(ynum, bet, rb);
        y[(i - 1) * KS] = yprev;
        gm[(i) * KS] = g;
        bet = cc + q * g;
        ynum = rhs + q * yprev;
        pm1 = p;
      };
      // (the votes as wave masks, one ballot per compare - each is the compare's own scalar result - so that the trip's
      // test is scalar arithmetic: a bool carried round the loop comes back as a VGPR and two more vector instructions)
      auto rare = [&]() { return __builtin_amdgcn_ballot_w64(tiny_nonzero(ynum)) | __builtin_amdgcn_ballot_w64(bet == 0.); };
      unsigned long long f_in = __builtin_amdgcn_ballot_w64(tiny_nonzero(ynum));   // of the state the next level starts from
      {   // two levels per trip; each half's operands are fetched while the other half runs
        int i = 2;
        double a_p = pb[(2) * KS], a_q = qq[(2) * KS], a_r = y[(2) * KS];
        for (; i + 1 <= nz; i += 2) {
          const double b_p = pb[(i + 1) * KS], b_q = qq[(i + 1) * KS], b_r = y[(i + 1) * KS];
          const double s_pm1 = pm1, s_bet = bet, s_ynum = ynum;
          level(i, a_p, a_q, a_r, std::false_type{});
          const unsigned long long f_mid = rare();
          level(i + 1, b_p, b_q, b_r, std::false_type{});
          if (__builtin_expect((f_in | f_mid) != 0ull, 0)) {
            pm1 = s_pm1; bet = s_bet; ynum = s_ynum;
            level(i, a_p, a_q, a_r, std::true_type{});
            level(i + 1, b_p, b_q, b_r, std::true_type{});
          }
          f_in = rare();
          a_p = pb[(i + 2) * KS]; a_q = qq[(i + 2) * KS]; a_r = y[(i + 2) * KS];   // (i + 2 <= nzp1: inside the rows; unused after the last trip)
        }
        if (i <= nz) {
          if (__builtin_expect(f_in != 0ull, 0)) level(i, a_p, a_q, a_r, std::true_type{});
          else level(i, a_p, a_q, a_r, std::false_type{});
        }
      }
      if (bet == 0.) { bad = 1; bet = 1.E-12; }
      y[(nz) * KS] = div_by_refined(ynum, bet, rcp_refine(bet));
      if (bad) sbad[sl * sbad_stride] = 1;
    }
  }
}

// ---- The same forward part with its LDS traffic issued by hand (default physics and optional physics; double diffusion
// keeps ps_thomas_uts_fwd: its S system's q is not seven rows behind its p).  What the compiler's two-level trip spends
// beside the recurrence - the votes as compares, or-s and a frexp per level, addresses, its own waits - is a third of
// its instructions (69 per two levels for 38 fp64 and two reciprocals).  Here a half-trip of two levels takes its
// operands - (p, q) of either level and the two right-hand sides: three ds_read2_b64 - from registers fetched while
// the half before ran (two register sets in turn: ps_backsub_from), carries the smallest |pivot| and the smallest
// exponent of a numerator along instead of comparing per level (one vote per half: a zero pivot, solvers.F90:140-151,
// or a tiny non-zero numerator), redoes the half on the IEEE path from its entry state and its operands - still in
// their registers - when the vote fires, and stores its four results with two ds_write2_b64.  Same operations on the
// same operands as ps_thomas_uts_fwd.
template <int XV>
__device__ __forceinline__ void ps_thomas_uts_fwd4(int W, double *slots, int SS, int nz, const int *sact, int sact_stride,
                                                   int *sbad, int sbad_stride, int lane)
{
  static_assert(XV != 2, "p and q rows seven apart");
  constexpr int KS = XV == 1 ? (int)Q_COUNT_EXT : (int)Q_COUNT;
  asm volatile("" : "+v"(lane));
  if (lane < 3 * W) {
    const int sl = lane / 3, sys = lane - 3 * sl;
    if (sact[sl * sact_stride]) {
      double *base = slots + sl * SS;
      const double *pb = base + ps_sysrows<XV>::p(sys), *qq = pb + 7;
      double *y = base + (Q_YU + sys), *gm = base + ps_sysrows<XV>::gam(sys);
      int bad = 0;
      double pm1 = pb[(1) * KS];
      double bet = 1. + pm1;
      double ynum = y[(1) * KS];
      asm volatile("" : "+v"(pm1), "+v"(bet), "+v"(ynum));   // their waits here, not inside the loop
      // one level, IEEE form, on operands at hand: gam(i) and y(i-1)
      auto slow_level = [&](double p, double q, double rhs, double &g_out, double &y_out) {
        if (bet == 0.) { bad = 1; bet = 1.E-12; }
        const double rb = rcp_refine(bet);
        const double g = div_by_refined(-pm1, bet, rb);
        const double yprev = div_by_refined(ynum, bet, rb);
        g_out = g; y_out = yprev;
        bet = ((1. + p) + q) + q * g;
        ynum = rhs + q * yprev;
        pm1 = p;
      };
      int i = 2;
      if (i + 1 <= nz) {
        // one address register per row (p with q seven rows behind it; the solution row, whose entry i-1 takes y(i-1) and
        // whose entries i, i+1, .. are the right-hand sides; gam), everything else is an offset field of the instruction:
        // H = the half-trip, counted from the levels worked on (0: levels i, i+1; 1: i+2, i+3; 2: the next trip's first)
        unsigned ap = ps_lds_addr(pb + i * KS), ay = ps_lds_addr(y + (i - 1) * KS), ag = ps_lds_addr(gm + i * KS);
        auto rd_pq = [&](unsigned a, auto H) { constexpr int o = 2 * KS * decltype(H)::value; return ps_lds_read2<o, o + 7>(a); };             // (p, q) of a level
        auto rd_pq1 = [&](unsigned a, auto H) { constexpr int o = 2 * KS * decltype(H)::value + KS; return ps_lds_read2<o, o + 7>(a); };      // ... of the next
        auto rd_rr = [&](unsigned a, auto H) { constexpr int o = 2 * KS * decltype(H)::value + KS; return ps_lds_read2<o, o + KS>(a); };       // right-hand sides of the two
        using H0 = std::integral_constant<int, 0>; using H1 = std::integral_constant<int, 1>; using H2 = std::integral_constant<int, 2>;
        // a half-trip at the level of (ay_, ag_) on operands (pq0, pq1, rr): two levels
        auto two = [&](auto H, const ps_d2 &pq0, const ps_d2 &pq1, const ps_d2 &rr) {
          const double s_pm1 = pm1, s_bet = bet, s_ynum = ynum;
          // the first level
          double amin = __builtin_fabs(bet);
          int emin = __builtin_amdgcn_frexp_exp(ynum);
          double rb = rcp_refine(bet);
          double g0 = div_fast(-pm1, bet, rb), y0 = div_fast(ynum, bet, rb);
          bet = ((1. + pq0.x) + pq0.y) + pq0.y * g0;
          ynum = rr.x + pq0.y * y0;
          // the second
          amin = __builtin_fmin(amin, __builtin_fabs(bet));
          emin = min(emin, __builtin_amdgcn_frexp_exp(ynum));
          rb = rcp_refine(bet);
          double g1 = div_fast(-pq0.x, bet, rb), y1 = div_fast(ynum, bet, rb);
          bet = ((1. + pq1.x) + pq1.y) + pq1.y * g1;
          ynum = rr.y + pq1.y * y1;
          pm1 = pq1.x;
          if (__builtin_expect(__builtin_amdgcn_ballot_w64(amin == 0. || emin < -960) != 0ull, 0)) {
            // again from the half's entry state: IEEE divisions, a zero pivot replaced (registers only: the stores below
            // are the only LDS traffic of a half, whichever way it went - the wait counts rely on it)
            pm1 = s_pm1; bet = s_bet; ynum = s_ynum;
            slow_level(pq0.x, pq0.y, rr.x, g0, y0);
            slow_level(pq1.x, pq1.y, rr.y, g1, y1);
          }
          // y(i-1), y(i); gam(i), gam(i+1)
          constexpr int o = 2 * KS * decltype(H)::value;
          ps_lds_write2<o, o + KS>(ay, y0, y1);
          ps_lds_write2<o, o + KS>(ag, g0, g1);
        };
        ps_d2 a0 = rd_pq(ap, H0{}), a1 = rd_pq1(ap, H0{}), ar = rd_rr(ay, H0{}), b0, b1, br;   // levels i, i+1
        ps_lds_wait<0>(a0, a1, ar);
        // Four levels per trip.  The second half's operands are fetched at the top of the trip and waited for behind the
        // first half; the next trip's first half is fetched behind the first half and waited for at the bottom of the
        // trip, behind the second: nothing is in flight across the loop's back edge (the compiler moves loop-carried
        // values between registers there - tools/check_inflight.py).  The fetches run two levels ahead of the levels
        // worked on: i + 5 <= nz + 2, the last entry of the rows.
        while (i + 3 <= nz) {
          b0 = rd_pq(ap, H1{}); b1 = rd_pq1(ap, H1{}); br = rd_rr(ay, H1{});
          two(H0{}, a0, a1, ar);
          a0 = rd_pq(ap, H2{}); a1 = rd_pq1(ap, H2{}); ar = rd_rr(ay, H2{});
          ps_lds_wait<5>(b0, b1, br);   // (behind it: the first half's two stores, the three reads just issued)
          two(H1{}, b0, b1, br);
          ps_lds_wait<2>(a0, a1, ar);   // (behind it: the second half's two stores)
          constexpr unsigned step = 4u * (unsigned)KS * 8u;
          i += 4; ap += step; ay += step; ag += step;
        }
        if (i + 1 <= nz) { two(H0{}, a0, a1, ar); i += 2; }
        ps_lds_drain();
      }
      // the level the halves leave over, and the last pivot
      if (i <= nz) {
        double g, yp;
        slow_level(pb[(i) * KS], qq[(i) * KS], y[(i) * KS], g, yp);
        y[(i - 1) * KS] = yp; gm[(i) * KS] = g;
      }
      if (bet == 0.) { bad = 1; bet = 1.E-12; }
      y[(nz) * KS] = div_by_refined(ynum, bet, rcp_refine(bet));
      if (bad) sbad[sl * sbad_stride] = 1;
    }
  }
}

template <int XV>
__device__ __forceinline__ void ps_thomas_uts_back(int W, double *slots, int SS, int nz, const int *sact, int sact_stride, int lane)
{
  constexpr int KS = XV == 2 ? (int)Q_COUNT_EXT_DD : XV == 1 ? (int)Q_COUNT_EXT : (int)Q_COUNT;
  asm volatile("" : "+v"(lane));
  if (lane < 3 * W) {
    const int sl = lane / 3, sys = lane - 3 * sl;
    if (sact[sl * sact_stride]) {
      double *base = slots + sl * SS;
      ps_backsub(base + (Q_YU + sys), base + ps_sysrows<XV>::gam(sys), KS, nz);
    }
  }
}

// ---- solver mode 1 (opt-in, mckpp_hip_set_solver_mode; NOT the reference's operation order - the oracle's
// orc_tridmat_2e restates it operation for operation): the same systems eliminated from both ends at once.  DIR = +1:
// the levels 1..m, m = nz/2, downward as above; DIR = -1: the levels nz..m+1 upward by the mirrored recurrence
//   g(i+1) = cu(i+1)/bet(i+1) = -q(i+1)/bet,  bet(i) = cc(i) - cl(i) g(i+1) = cc(i) + p(i) g(i+1),
//   z(i) = (rhs(i) - cl(i) z(i+1))/bet(i) = (rhs(i) + p(i) z(i+1))/bet(i)
// on another wave; each is a chain of nz/2 steps.  The upper half leaves z(1..m-1) in the solution row, z(m) at its
// index 0, gam(2..m) in the gam row and gam(m+1) at its index 0; the lower half z(m+2..nz), z(m+1) at index nz+2 of
// the solution row, g(m+2..nz) and g(m+1) at index 1 of the gam row (those entries of the rows are free by now:
// nothing there for the other half's reads to race with).
template <int XV, int DIR>
__device__ __forceinline__ void ps_thomas2_uts_fwd(int W, double *slots, int SS, int nz, const int *sact, int sact_stride,
                                                   int *sbad, int sbad_stride, int lane)
{
  constexpr int KS = XV == 2 ? (int)Q_COUNT_EXT_DD : XV == 1 ? (int)Q_COUNT_EXT : (int)Q_COUNT;
  // (opaque: what is derived from the lane number is formed here, in every pass, instead of living in registers
  // across the whole persistent loop - the compiler would hoist it out, and the kernel has no VGPR to spare)
  asm volatile("" : "+v"(lane));
  if (lane < 3 * W) {
    const int sl = lane / 3, sys = lane - 3 * sl;
    if (sact[sl * sact_stride]) {
      double *base = slots + sl * SS;
      const double *pb = base + ps_sysrows<XV>::p(sys), *qq = XV == 2 ? base + ps_sysrows<XV>::q(sys) : pb + 7;
      double *y = base + (Q_YU + sys), *gm = base + ps_sysrows<XV>::gam(sys);
      const int m = nz >> 1;
      const int i0 = DIR > 0 ? 1 : nz;
      int bad = 0;
      // carry: p(i-1) on the way down (cl(i-1) = -p(i-1)), q(i+1) on the way up (cu(i+1) = -q(i+1))
      double carry = DIR > 0 ? pb[(1) * KS] : qq[(nz) * KS];
      double bet = DIR > 0 ? 1. + carry : (1. + pb[(nz) * KS]) + carry;   // cc(1) | cc(nz)
      if (DIR < 0 && bet == 0.) { bad = 1; bet = 1.E-12; }   // (cc(nz) is a pivot tridmat checks; cc(1) it does not)
      double ynum = y[(i0) * KS];
      auto level = [&](int i, double p, double q, double rhs, auto slow) {
        if (slow.value && bet == 0.) { bad = 1; bet = 1.E-12; }
        const double cm1 = -carry;
        const double cc = (1. + p) + q;
        const double rb = rcp_refine(bet);
        const double g = slow.value ? div_by_refined(cm1, bet, rb) : div_fast(cm1, bet, rb);
        const double yprev = slow.value ? div_by_refined(ynum, bet, rb) : div_fast(ynum, bet, rb);
        y[(i - DIR) * KS] = yprev;
        gm[(DIR > 0 ? i : i + 1) * KS] = g;
        const double mult = DIR > 0 ? q : p;
        bet = cc + mult * g;
        ynum = rhs + mult * yprev;
        carry = DIR > 0 ? p : q;
      };
      auto rare = [&]() { return __builtin_amdgcn_ballot_w64(tiny_nonzero(ynum)) | __builtin_amdgcn_ballot_w64(bet == 0.); };   // (as ps_thomas_uts_fwd)
      unsigned long long f_in = __builtin_amdgcn_ballot_w64(tiny_nonzero(ynum));
      {   // two levels per trip; each half's operands are fetched while the other half runs
        int i = i0 + DIR, left = (DIR > 0 ? m : nz - m) - 1;
        double a_p = 0., a_q = 0., a_r = 0.;
        if (left > 0) { a_p = pb[(i) * KS]; a_q = qq[(i) * KS]; a_r = y[(i) * KS]; }
        for (; left >= 2; left -= 2, i += 2 * DIR) {
          const double b_p = pb[(i + DIR) * KS], b_q = qq[(i + DIR) * KS], b_r = y[(i + DIR) * KS];
          const double s_carry = carry, s_bet = bet, s_ynum = ynum;
          level(i, a_p, a_q, a_r, std::false_type{});
          const unsigned long long f_mid = rare();
          level(i + DIR, b_p, b_q, b_r, std::false_type{});
          if (__builtin_expect((f_in | f_mid) != 0ull, 0)) {
            carry = s_carry; bet = s_bet; ynum = s_ynum;
            level(i, a_p, a_q, a_r, std::true_type{});
            level(i + DIR, b_p, b_q, b_r, std::true_type{});
          }
          f_in = rare();
          if (left >= 3) { a_p = pb[(i + 2 * DIR) * KS]; a_q = qq[(i + 2 * DIR) * KS]; a_r = y[(i + 2 * DIR) * KS]; }
        }
        if (left == 1) {
          if (__builtin_expect(f_in != 0ull, 0)) level(i, a_p, a_q, a_r, std::true_type{});
          else level(i, a_p, a_q, a_r, std::false_type{});
        }
      }
      if (bet == 0.) { bad = 1; bet = 1.E-12; }
      const double rb = rcp_refine(bet);
      // (not into y(m), y(m+1): each of those is written by one wave while the other may still have to read it)
      y[(DIR > 0 ? 0 : nz + 2) * KS] = div_by_refined(ynum, bet, rb);        // z(m) | z(m+1) at the free ends of the row
      gm[(DIR > 0 ? 0 : 1) * KS] = div_by_refined(-carry, bet, rb);          // gam(m+1) | g(m+1)
      if (bad) sbad[sl * sbad_stride] = 1;
    }
  }
}

// The same with its LDS traffic issued by hand, as ps_thomas_uts_fwd4 above (default and optional physics): half-trips of
// two levels on two register sets, one vote per half, the half redone on the IEEE path from its entry state when the vote
// fires, results stored with two ds_write2_b64.  A half's addresses are those of the LOWER of its two levels (lo): the
// way down it works lo then lo + 1, the way up lo + 1 then lo; the three address registers move by two levels per half in
// the sweep's direction, every other displacement is an offset field.  Same operations on the same operands as
// ps_thomas2_uts_fwd.
template <int XV, int DIR>
__device__ __forceinline__ void ps_thomas2_uts_fwd4(int W, double *slots, int SS, int nz, const int *sact, int sact_stride,
                                                    int *sbad, int sbad_stride, int lane)
{
  static_assert(XV != 2, "p and q rows seven apart");
  constexpr int KS = XV == 1 ? (int)Q_COUNT_EXT : (int)Q_COUNT;
  asm volatile("" : "+v"(lane));
  if (lane < 3 * W) {
    const int sl = lane / 3, sys = lane - 3 * sl;
    if (sact[sl * sact_stride]) {
      double *base = slots + sl * SS;
      const double *pb = base + ps_sysrows<XV>::p(sys), *qq = pb + 7;
      double *y = base + (Q_YU + sys), *gm = base + ps_sysrows<XV>::gam(sys);
      const int m = nz >> 1;
      const int i0 = DIR > 0 ? 1 : nz;
      int bad = 0;
      double carry = DIR > 0 ? pb[(1) * KS] : qq[(nz) * KS];
      double bet = DIR > 0 ? 1. + carry : (1. + pb[(nz) * KS]) + carry;   // cc(1) | cc(nz)
      if (DIR < 0 && bet == 0.) { bad = 1; bet = 1.E-12; }
      double ynum = y[(i0) * KS];
      asm volatile("" : "+v"(carry), "+v"(bet), "+v"(ynum));   // their waits here, not inside the loop
      auto slow_level = [&](double p, double q, double rhs, double &g_out, double &y_out) {
        if (bet == 0.) { bad = 1; bet = 1.E-12; }
        const double rb = rcp_refine(bet);
        const double g = div_by_refined(-carry, bet, rb);
        const double yprev = div_by_refined(ynum, bet, rb);
        g_out = g; y_out = yprev;
        const double mult = DIR > 0 ? q : p;
        bet = ((1. + p) + q) + mult * g;
        ynum = rhs + mult * yprev;
        carry = DIR > 0 ? p : q;
      };
      int i = i0 + DIR, left = (DIR > 0 ? m : nz - m) - 1;   // the next level, and how many there are
      if (left >= 2) {
        // ap: (p, q) of level lo; ay: the solution row at lo - 1 (down) | lo (up); ag: gam at lo (down) | lo + 1 (up)
        const int lo = DIR > 0 ? i : i - 1;
        unsigned ap = ps_lds_addr(pb + lo * KS), ay = ps_lds_addr(y + (DIR > 0 ? lo - 1 : lo) * KS), ag = ps_lds_addr(gm + (DIR > 0 ? lo : lo + 1) * KS);
        const unsigned hstep = (unsigned)(DIR * 2 * KS * 8);   // a half further (two levels, in bytes; wraps as it should the way up)
        auto rd_lo = [&](unsigned a) { return ps_lds_read2<0, 7>(a); };             // (p, q) of lo
        auto rd_hi = [&](unsigned a) { return ps_lds_read2<KS, KS + 7>(a); };       // ... of lo + 1
        auto rd_rr = [&](unsigned a) { return DIR > 0 ? ps_lds_read2<KS, 2 * KS>(a) : ps_lds_read2<0, KS>(a); };   // right-hand sides of lo, lo + 1
        // a half at (ay_, ag_) on the operands of its two levels
        auto two = [&](unsigned ay_, unsigned ag_, const ps_d2 &pq_lo, const ps_d2 &pq_hi, const ps_d2 &rr) {
          const ps_d2 &pq0 = DIR > 0 ? pq_lo : pq_hi, &pq1 = DIR > 0 ? pq_hi : pq_lo;   // in the order they are worked
          const double r0 = DIR > 0 ? rr.x : rr.y, r1 = DIR > 0 ? rr.y : rr.x;
          const double s_carry = carry, s_bet = bet, s_ynum = ynum;
          double amin = __builtin_fabs(bet);
          int emin = __builtin_amdgcn_frexp_exp(ynum);
          double rb = rcp_refine(bet);
          double g0 = div_fast(-carry, bet, rb), y0 = div_fast(ynum, bet, rb);
          const double m0 = DIR > 0 ? pq0.y : pq0.x;
          bet = ((1. + pq0.x) + pq0.y) + m0 * g0;
          ynum = r0 + m0 * y0;
          amin = __builtin_fmin(amin, __builtin_fabs(bet));
          emin = min(emin, __builtin_amdgcn_frexp_exp(ynum));
          rb = rcp_refine(bet);
          double g1 = div_fast(DIR > 0 ? -pq0.x : -pq0.y, bet, rb), y1 = div_fast(ynum, bet, rb);
          const double m1 = DIR > 0 ? pq1.y : pq1.x;
          bet = ((1. + pq1.x) + pq1.y) + m1 * g1;
          ynum = r1 + m1 * y1;
          carry = DIR > 0 ? pq1.x : pq1.y;
          if (__builtin_expect(__builtin_amdgcn_ballot_w64(amin == 0. || emin < -960) != 0ull, 0)) {
            carry = s_carry; bet = s_bet; ynum = s_ynum;
            slow_level(pq0.x, pq0.y, r0, g0, y0);
            slow_level(pq1.x, pq1.y, r1, g1, y1);
          }
          // down: y(lo-1), y(lo) and gam(lo), gam(lo+1); up: y(lo+2) <- the first level's, y(lo+1) and gam(lo+2), gam(lo+1)
          if (DIR > 0) { ps_lds_write2<0, KS>(ay_, y0, y1); ps_lds_write2<0, KS>(ag_, g0, g1); }
          else { ps_lds_write2<KS, 2 * KS>(ay_, y1, y0); ps_lds_write2<0, KS>(ag_, g1, g0); }
        };
        ps_d2 a0 = rd_lo(ap), a1 = rd_hi(ap), ar = rd_rr(ay), b0, b1, br;
        ps_lds_wait<0>(a0, a1, ar);
        // Four levels per trip (see ps_thomas_uts_fwd4): the second half's operands fetched at the top and waited for
        // behind the first half, the next trip's first half fetched behind the first half and waited for at the bottom.
        // The fetches run two levels ahead of the levels worked on - inside the rows either way (levels >= m - 3 > 0 the
        // way up, <= m + 4 the way down; nz >= 16 here).
        while (left >= 4) {
          const unsigned apb = ap + hstep, ayb = ay + hstep, agb = ag + hstep;
          b0 = rd_lo(apb); b1 = rd_hi(apb); br = rd_rr(ayb);
          two(ay, ag, a0, a1, ar);
          ap = apb + hstep; ay = ayb + hstep; ag = agb + hstep;
          a0 = rd_lo(ap); a1 = rd_hi(ap); ar = rd_rr(ay);
          ps_lds_wait<5>(b0, b1, br);   // (behind it: the first half's two stores, the three reads just issued)
          two(ayb, agb, b0, b1, br);
          ps_lds_wait<2>(a0, a1, ar);   // (behind it: the second half's two stores)
          left -= 4; i += 4 * DIR;
        }
        if (left >= 2) { two(ay, ag, a0, a1, ar); left -= 2; i += 2 * DIR; }
        ps_lds_drain();
      }
      if (left == 1) {   // the level the halves leave over
        double g, yp;
        slow_level(pb[(i) * KS], qq[(i) * KS], y[(i) * KS], g, yp);
        y[(i - DIR) * KS] = yp; gm[(DIR > 0 ? i : i + 1) * KS] = g;
      }
      if (bet == 0.) { bad = 1; bet = 1.E-12; }
      const double rb = rcp_refine(bet);
      y[(DIR > 0 ? 0 : nz + 2) * KS] = div_by_refined(ynum, bet, rb);        // z(m) | z(m+1) at the free ends of the row
      gm[(DIR > 0 ? 0 : 1) * KS] = div_by_refined(-carry, bet, rb);          // gam(m+1) | g(m+1)
      if (bad) sbad[sl * sbad_stride] = 1;
    }
  }
}

// ... then (after a barrier) the two unknowns in the middle from their 2x2 system, by both waves alike,
//   y(m) = (z(m) - gam(m+1) z(m+1)) / (1 - gam(m+1) g(m+1)),  y(m+1) = z(m+1) - g(m+1) y(m),
// and the two substitutions away from it: DIR = +1 upwards from y(m), DIR = -1 downwards from y(m+1).
template <int XV, int DIR>
__device__ __forceinline__ void ps_thomas2_uts_back(int W, double *slots, int SS, int nz, const int *sact, int sact_stride,
                                                    int *sbad, int sbad_stride, int lane)
{
  constexpr int KS = XV == 2 ? (int)Q_COUNT_EXT_DD : XV == 1 ? (int)Q_COUNT_EXT : (int)Q_COUNT;
  asm volatile("" : "+v"(lane));   // (see ps_thomas2_uts_fwd)
  if (lane < 3 * W) {
    const int sl = lane / 3, sys = lane - 3 * sl;
    if (sact[sl * sact_stride]) {
      double *base = slots + sl * SS;
      double *y = base + (Q_YU + sys), *gm = base + ps_sysrows<XV>::gam(sys);
      const int m = nz >> 1;
      const double zt = y[0], zb = y[(nz + 2) * KS], gt = gm[0], gb = gm[KS];
      double den = 1. - gt * gb;
      if (__builtin_expect(den == 0., 0)) { den = 1.E-12; sbad[sl * sbad_stride] = 1; }   // the middle system's pivot, like tridmat's (both waves alike)
      const double ym = (zt - gt * zb) / den;
      if (DIR > 0) {
        y[(m) * KS] = ym;
        ps_backsub_from(y, gm, KS, m, ym);
      } else {
        const double ym1 = zb - gb * ym;
        y[(m + 1) * KS] = ym1;
        ps_fwdsub_from(y, gm, KS, m + 2, nz, ym1);
      }
    }
  }
}

// V on the stored momentum factorisation, forward part; lane = slot.  The U sweep has kept q(i) = tri(i,0)
// difm(i-1) = -cu(i) (row Q_GM); L7 (all threads) has formed the pivots again and their refined reciprocals
// (rows Q_BET and Q_DT, dead once T and S are solved) - so a level is n = rhs + q y (= rhs - cu y, ocnint /
// solvers.F90:153) and one div_fast.  A tiny non-zero numerator
// (which div_fast must not see) is looked for once per trip of two levels, after the fact; the trip is then
// redone from the value it started with, with IEEE divisions.  118 cycles per level alone on a CU (150 with
// compiler-scheduled reads of single values; tools/ubench/sweeps.hip).
__device__ __forceinline__ void ps_thomas_v_fwd(int W, double *slots, int SS, int KS, int nz, const int *sact,
                                                int sact_stride, int lane)
{
  asm volatile("" : "+v"(lane));
  if (lane < W && sact[lane * sact_stride]) {
    double *base = slots + lane * SS;
    const double *betm = base + Q_BET, *rbm = base + Q_DT, *qm = base + Q_GM;
    double *y = base + Q_YV;
    const double b1 = betm[(1) * KS];
    double yy = div_by_refined(y[(1) * KS], b1, rbm[(1) * KS]);
    y[(1) * KS] = yy;
    asm volatile("" : "+v"(yy));   // the waits for its operands here, not inside the loop below
    int i = 2;
    // A trip is two levels; its eight operands - (1/bet, q) and (rhs, bet) of either level, all in the slot's
    // level-interleaved block - come with four ds_read2_b64 issued one trip ahead (ps_lds_read2 / ps_lds_wait, above).
    if (i + 1 <= nz) {
      static_assert(Q_DT == 1 && Q_YV == 6 && Q_GM == 7 && Q_BET == 8, "offsets of the V sweep's operands in a level block");
      const unsigned step = 2u * (unsigned)KS * 8u;
      unsigned ad = ps_lds_addr(base + i * KS);
      // rq = (1/bet, q), hb = (rhs, bet)
      auto rd_qr0 = [&](unsigned a) { return ps_lds_read2<1, 7>(a); };
      auto rd_hb0 = [&](unsigned a) { return ps_lds_read2<6, 8>(a); };
      auto rd_qr1 = [&](unsigned a) { return KS == 9 ? ps_lds_read2<10, 16>(a) : KS == 11 ? ps_lds_read2<12, 18>(a) : ps_lds_read2<16, 22>(a); };
      auto rd_hb1 = [&](unsigned a) { return KS == 9 ? ps_lds_read2<15, 17>(a) : KS == 11 ? ps_lds_read2<17, 19>(a) : ps_lds_read2<21, 23>(a); };
      auto body = [&](unsigned aw, const ps_d2 &rq0, const ps_d2 &hb0, const ps_d2 &rq1, const ps_d2 &hb1) {
        const double y_in = yy;
        const double n0 = hb0.x + rq0.y * yy;
        double y0 = div_fast(n0, hb0.y, rq0.x);
        const double n1 = hb1.x + rq1.y * y0;
        double y1 = div_fast(n1, hb1.y, rq1.x);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(tiny_nonzero(n0) || tiny_nonzero(n1)) != 0ull, 0)) {
          y0 = (hb0.x + rq0.y * y_in) / hb0.y;
          y1 = (hb1.x + rq1.y * y0) / hb1.y;
        }
        yy = y1;
        if (KS == 9) ps_lds_write2<6, 15>(aw, y0, y1);
        else if (KS == 11) ps_lds_write2<6, 17>(aw, y0, y1);
        else ps_lds_write2<6, 21>(aw, y0, y1);
      };
      // two levels on div_fast; says whether either numerator was one that div_fast must not see
      auto fast2 = [&](const ps_d2 &rq0, const ps_d2 &hb0, const ps_d2 &rq1, const ps_d2 &hb1, double &y0, double &y1) -> bool {
        const double n0 = hb0.x + rq0.y * yy;
        y0 = div_fast(n0, hb0.y, rq0.x);
        const double n1 = hb1.x + rq1.y * y0;
        y1 = div_fast(n1, hb1.y, rq1.x);
        yy = y1;
        return tiny_nonzero(n0) || tiny_nonzero(n1);
      };
      auto store2 = [&](unsigned aw, double y0, double y1) {
        if (KS == 9) ps_lds_write2<6, 15>(aw, y0, y1);
        else if (KS == 11) ps_lds_write2<6, 17>(aw, y0, y1);
        else ps_lds_write2<6, 21>(aw, y0, y1);
      };
      ps_d2 a0 = rd_qr0(ad), a1 = rd_hb0(ad), a2 = rd_qr1(ad), a3 = rd_hb1(ad), b0, b1_, b2, b3;
      while (i + 5 <= nz) {   // this trip, the next, and one more after it; one vote on the numerators per two trips
        b0 = rd_qr0(ad + step); b1_ = rd_hb0(ad + step); b2 = rd_qr1(ad + step); b3 = rd_hb1(ad + step);
        ps_lds_wait<4>(a0, a1, a2, a3);
        const double y_in = yy, rhs0 = a1.x, rhs1 = a3.x;   // what a redo of the first trip needs (its solution replaces rhs)
        double ya0, ya1, yb0, yb1;
        const bool fa = fast2(a0, a1, a2, a3, ya0, ya1);
        store2(ad, ya0, ya1);
        a0 = rd_qr0(ad + 2 * step); a1 = rd_hb0(ad + 2 * step); a2 = rd_qr1(ad + 2 * step); a3 = rd_hb1(ad + 2 * step);
        ps_lds_wait<4>(b0, b1_, b2, b3);
        const bool fb = fast2(b0, b1_, b2, b3, yb0, yb1);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(fa || fb) != 0ull, 0)) {   // the four levels again, IEEE divisions
          ya0 = (rhs0 + qm[(i) * KS] * y_in) / betm[(i) * KS];
          ya1 = (rhs1 + qm[(i + 1) * KS] * ya0) / betm[(i + 1) * KS];
          y[(i) * KS] = ya0; y[(i + 1) * KS] = ya1;
          yb0 = (b1_.x + b0.y * ya1) / b1_.y;
          yb1 = (b3.x + b2.y * yb0) / b3.y;
          yy = yb1;
        }
        store2(ad + step, yb0, yb1);
        i += 4; ad += 2 * step;
      }
      // (past the loop nothing stays in flight across a branch: see ps_backsub)
      ps_lds_wait<0>(a0, a1, a2, a3);
      body(ad, a0, a1, a2, a3);
      i += 2;
      if (i + 1 <= nz) {
        b0 = rd_qr0(ad + step); b1_ = rd_hb0(ad + step); b2 = rd_qr1(ad + step); b3 = rd_hb1(ad + step);
        ps_lds_wait<0>(b0, b1_, b2, b3);
        body(ad + step, b0, b1_, b2, b3);
        i += 2;
      }
      ps_lds_drain();
    }
    if (i <= nz) {
      const double n0 = y[(i) * KS] + qm[(i) * KS] * yy;
      yy = div_fast_guarded(n0, betm[(i) * KS], rbm[(i) * KS]);
      y[(i) * KS] = yy;
    }
  }
}

__device__ __forceinline__ void ps_thomas_v_back(int W, double *slots, int SS, int KS, int gam_row, int nz, const int *sact,
                                                 int sact_stride, int lane)
{
  asm volatile("" : "+v"(lane));
  if (lane < W && sact[lane * sact_stride]) {
    double *base = slots + lane * SS;
    ps_backsub(base + Q_YV, base + gam_row, KS, nz);
  }
}

// Solver mode 1, V: both halves of the two-ended elimination on ONE wave - lane s works downward from level 1 of
// slot s, lane 32+s upward from its level nz - on what L7 has laid out per level for either direction alike: the
// pivot of the level (row Q_BET) and its refined reciprocal (Q_DT), the multiplier of the neighbour's solution in
// the numerator (Q_GM: q(k) in the upper half, p(k) in the lower one) and the multiplier of the substitution (Q_DM:
// gam(k+1) | g(k)).  So the two directions differ in their start level and the sign of their address step only.
template <int OFF>
__device__ __forceinline__ void ps_lds_write1(unsigned addr, double a)
{
  asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(addr), "v"(a), "n"(OFF) : "memory");
}
__device__ __forceinline__ void ps_thomas2_v(int W, double *slots, int SS, int KS, int gam_row, int nz, const int *sact,
                                             int sact_stride, int *sbad, int sbad_stride, int lane)
{
  static_assert(Q_DM == 0 && Q_DT == 1 && Q_YV == 6 && Q_GM == 7 && Q_BET == 8, "offsets of the V sweep's operands in a level block");
  asm volatile("" : "+v"(lane));   // (see ps_thomas2_uts_fwd)
  const int sl = lane & 31;
  const bool up = lane >= 32;   // the lower half, worked upward
  if (sl < W && sact[sl * sact_stride]) {
    double *base = slots + sl * SS;
    const int m = nz >> 1;
    const int i0 = up ? nz : 1, iend = up ? m + 1 : m;
    const int dstep = up ? -KS * 8 : KS * 8;   // bytes from a level to the next one of this lane's sweep
    int left = (up ? nz - m : m) - 1;          // levels after the first
    const int common = m - 1;                  // ... of either direction (the lower half may have one more)
    double yy;
    {
      const double *l0 = base + i0 * KS;
      yy = div_by_refined(l0[Q_YV], l0[Q_BET], l0[Q_DT]);
      base[i0 * KS + Q_YV] = yy;
    }
    asm volatile("" : "+v"(yy));
    unsigned ad = ps_lds_addr(base + i0 * KS) + (unsigned)dstep;   // the level to enter next
    // rq = (1/bet, multiplier), hb = (rhs, bet) of a level
    auto rd_rq = [&](unsigned a) { return ps_lds_read2<1, 7>(a); };
    auto rd_hb = [&](unsigned a) { return ps_lds_read2<6, 8>(a); };
    auto two = [&](unsigned a0_, unsigned a1_, const ps_d2 &rq0, const ps_d2 &hb0, const ps_d2 &rq1, const ps_d2 &hb1) {
      const double y_in = yy;
      const double n0 = hb0.x + rq0.y * yy;
      double y0 = div_fast(n0, hb0.y, rq0.x);
      const double n1 = hb1.x + rq1.y * y0;
      double y1 = div_fast(n1, hb1.y, rq1.x);
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(tiny_nonzero(n0) || tiny_nonzero(n1)) != 0ull, 0)) {
        y0 = (hb0.x + rq0.y * y_in) / hb0.y;
        y1 = (hb1.x + rq1.y * y0) / hb1.y;
      }
      yy = y1;
      ps_lds_write1<48>(a0_, y0);
      ps_lds_write1<48>(a1_, y1);
    };
    int done = 0;
    if (common >= 2) {   // trips of two levels, the next trip's operands in flight (ps_lds_read2 / ps_lds_wait, above)
      ps_d2 a0 = rd_rq(ad), a1 = rd_hb(ad), a2 = rd_rq(ad + dstep), a3 = rd_hb(ad + dstep), b0, b1, b2, b3;
      while (done + 6 <= common) {   // this trip, the next, and one more after it
        const unsigned adb = ad + 2 * dstep, adc = ad + 4 * dstep;
        b0 = rd_rq(adb); b1 = rd_hb(adb); b2 = rd_rq(adb + dstep); b3 = rd_hb(adb + dstep);
        ps_lds_wait<4>(a0, a1, a2, a3);
        two(ad, ad + dstep, a0, a1, a2, a3);
        a0 = rd_rq(adc); a1 = rd_hb(adc); a2 = rd_rq(adc + dstep); a3 = rd_hb(adc + dstep);
        ps_lds_wait<4>(b0, b1, b2, b3);
        two(adb, adb + dstep, b0, b1, b2, b3);
        done += 4; ad = adc;
      }
      // (past the loop nothing stays in flight across a branch: see ps_backsub_from)
      ps_lds_wait<0>(a0, a1, a2, a3);
      two(ad, ad + dstep, a0, a1, a2, a3);
      done += 2; ad += 2 * dstep;
      if (done + 2 <= common) {
        b0 = rd_rq(ad); b1 = rd_hb(ad); b2 = rd_rq(ad + dstep); b3 = rd_hb(ad + dstep);
        ps_lds_wait<0>(b0, b1, b2, b3);
        two(ad, ad + dstep, b0, b1, b2, b3);
        done += 2; ad += 2 * dstep;
      }
      ps_lds_drain();
    }
    left -= done;
    while (__builtin_amdgcn_ballot_w64(left > 0) != 0ull) {   // what is left of either direction: at most two levels
      if (left > 0) {
        const double *l = reinterpret_cast<const double *>(base) + ((int)(ad - ps_lds_addr(base)) >> 3);
        const double n0 = l[Q_YV] + l[Q_GM] * yy;
        yy = div_fast_guarded(n0, l[Q_BET], l[Q_DT]);
        const_cast<double *>(l)[Q_YV] = yy;
        ad += dstep;
      }
      --left;
    }
    // the two unknowns in the middle: y(m) = (z(m) - gam(m+1) z(m+1)) / (1 - gam(m+1) g(m+1)), y(m+1) = z(m+1) - g(m+1) y(m)
    const double other = __shfl_xor(yy, 32);
    const double zt = up ? other : yy, zb = up ? yy : other;
    const double gt = base[gam_row], gb = base[KS + gam_row];
    double den = 1. - gt * gb;
    if (__builtin_expect(den == 0., 0)) { den = 1.E-12; sbad[sl * sbad_stride] = 1; }   // (as in ps_thomas2_uts_back)
    const double ym = (zt - gt * zb) / den;
    const double ym1 = zb - gb * ym;
    yy = up ? ym1 : ym;
    base[iend * KS + Q_YV] = yy;
    asm volatile("" : "+v"(yy));
    // substitution away from the middle: y(k) = z(k) - mult(k) y(previous), mult in row Q_DM (L7)
    const int bstep = -dstep;
    int bleft = up ? nz - m - 1 : m - 1;
    unsigned ab = ps_lds_addr(base + iend * KS) + (unsigned)bstep;
    auto rd_mz = [&](unsigned a) { return ps_lds_read2<0, 6>(a); };   // (mult, z)
    auto four = [&](unsigned a_, const ps_d2 &v0, const ps_d2 &v1, const ps_d2 &v2, const ps_d2 &v3) {
      yy = v0.y - v0.x * yy; ps_lds_write1<48>(a_, yy);
      yy = v1.y - v1.x * yy; ps_lds_write1<48>(a_ + bstep, yy);
      yy = v2.y - v2.x * yy; ps_lds_write1<48>(a_ + 2 * bstep, yy);
      yy = v3.y - v3.x * yy; ps_lds_write1<48>(a_ + 3 * bstep, yy);
    };
    int bdone = 0;
    if (common >= 4) {
      ps_d2 a0 = rd_mz(ab), a1 = rd_mz(ab + bstep), a2 = rd_mz(ab + 2 * bstep), a3 = rd_mz(ab + 3 * bstep), b0, b1, b2, b3;
      while (bdone + 12 <= common) {
        const unsigned abb = ab + 4 * bstep, abc = ab + 8 * bstep;
        b0 = rd_mz(abb); b1 = rd_mz(abb + bstep); b2 = rd_mz(abb + 2 * bstep); b3 = rd_mz(abb + 3 * bstep);
        ps_lds_wait<4>(a0, a1, a2, a3);
        four(ab, a0, a1, a2, a3);
        a0 = rd_mz(abc); a1 = rd_mz(abc + bstep); a2 = rd_mz(abc + 2 * bstep); a3 = rd_mz(abc + 3 * bstep);
        ps_lds_wait<4>(b0, b1, b2, b3);
        four(abb, b0, b1, b2, b3);
        bdone += 8; ab = abc;
      }
      ps_lds_wait<0>(a0, a1, a2, a3);
      four(ab, a0, a1, a2, a3);
      bdone += 4; ab += 4 * bstep;
      if (bdone + 4 <= common) {
        b0 = rd_mz(ab); b1 = rd_mz(ab + bstep); b2 = rd_mz(ab + 2 * bstep); b3 = rd_mz(ab + 3 * bstep);
        ps_lds_wait<0>(b0, b1, b2, b3);
        four(ab, b0, b1, b2, b3);
        bdone += 4; ab += 4 * bstep;
      }
      ps_lds_drain();
    }
    bleft -= bdone;
    while (__builtin_amdgcn_ballot_w64(bleft > 0) != 0ull) {   // at most four levels
      if (bleft > 0) {
        double *l = base + ((int)(ab - ps_lds_addr(base)) >> 3);
        yy = l[Q_YV] - l[Q_DM] * yy;
        l[Q_YV] = yy;
        ab += bstep;
      }
      --bleft;
    }
  }
}

#ifndef MCKPP_PS_FWD4
#define MCKPP_PS_FWD4 1
#endif
constexpr bool PS_FWD4 = MCKPP_PS_FWD4 != 0;   // the U,T,S forward sweep four levels per trip (ps_thomas_uts_fwd4); 0: the two-level form (A/B builds)

template <int KS>
struct strided {   // x[i] of a level-interleaved row
  double *b;
  __device__ __forceinline__ double &operator[](int i) const { return b[i * KS]; }
};

// XV: physics variant (rows above); SM: tridiagonal solver mode - 0 the reference's order of operations
// (solvers.F90:112-161), 1 the two-ended elimination (opt-in, ps_thomas2_*)
// LF: 0, or the number of level items per column (nzp1 + 2) as a compile-time constant - the item arithmetic of the level
// phases (item -> slot, level; the rows' strides) with literals instead of uniform values held in (spilled) SGPRs
// (the slot count and the slot stride as literals too: measured, -1 % at 60 levels, not kept)
template <int XV, int SM, int LF = 0>
#ifndef MCKPP_PS_MINW
#define MCKPP_PS_MINW 4
#endif
__global__ __launch_bounds__(1024, MCKPP_PS_MINW) void k_column_ps(const mckpp_kparams *__restrict__ pp, const int ntime, const int L_arg,
                                                     const int W0, const unsigned Lmagic_arg, const int SS /* ps_ss(L, XV, W0) */)
{
  const int L = LF > 0 ? LF : L_arg;
  const unsigned Lmagic = LF > 0 ? (unsigned)(0x100000000ull / (unsigned long long)(LF > 0 ? LF : 1)) + 1u : Lmagic_arg;
  // through the block typed with global pointers (mckpp_device.h): global_load / global_store, SGPR bases
  const mckpp_kparams_dev &p = *reinterpret_cast<const mckpp_kparams_dev *>(pp);
  extern __shared__ double lds[];
  constexpr bool EXT = XV != 0, DD = XV == 2;
  constexpr int ROWS = XV == 2 ? (int)Q_COUNT_EXT_DD : XV == 1 ? (int)Q_COUNT_EXT : (int)Q_COUNT;
  const int NL = ps_nl(L);
  const int tid = threadIdx.x, lane = tid & 63, lane_k = lane;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nz = LF > 0 ? LF - 3 : p.nz, nzp1 = LF > 0 ? LF - 2 : p.nzp1;   // (L = nzp1 + 2)
  double *cst = lds;
  const strided<K_STRIDE> c_zm{cst + K_ZM}, c_hm{cst + K_HM}, c_t0{cst + K_T0}, c_t1{cst + K_T1}, c_rdz{cst + K_RDZ},
      c_dtohk{cst + K_DTOHK};
  double *c_misc = lds + K_STRIDE * NL;
  double *const slots0 = c_misc + 2;
  double *const screc0 = slots0 + W0 * SS;
  int *const sirec0 = reinterpret_cast<int *>(screc0 + W0 * C_COUNT);
  // [0] some slot active, [1] some slot finishing, [2] the level the bulk-Ri scan of this pass ended at, [3] the level
  // down to which L3 forms the bulk Richardson numbers (a guess from the pass before), [4] the guess was too shallow,
  // [5] the scan stopped early, [6] some slot waits for its ticket, [7] the queue M0 draws from,
  // [S_KVIEW] the number of slots of the view the workgroup works in (0: every slot), [S_GOVIEW] the number of slots of
  // the view to go on in from this pass's L2 on (0: no change), [S_DRAIN] M0 does not refill for now (stragglers are being
  // left alone, or the queue is used up), [S_MAP ..] the slots of the view
  enum { S_KVIEW = 8, S_DRAIN = 9, S_GOVIEW = 10, S_MAP = 16, S_MAPMAX = 16 };
  int *s_flags = sirec0 + W0 * I_COUNT;
  // ---- the VIEW a pass works in.  Normally every slot of the workgroup, the items of the first trip dealt from thread 0
  // on.  A column that iterates towards itermax (200 passes where the others take 6) bounds a long run by its own chain
  // of passes - and such columns come in bands of neighbours, which consecutive tickets hand to the slots of one
  // workgroup.  So a workgroup that holds some lets its other slots run empty (M0) and then goes on in a view of the k
  // slots that are left: their items - k L of them, one per thread - dealt to the waves other than the manager's, which
  // has none.  Everything the full workgroup overlaps with the manager's serial phases (M1 | L2, the right-hand side of U
  // under M3, the next L1 under the V sweep, the control under L6) then overlaps for these columns too - with their items
  // on the manager wave itself, or spread over two trips among the items of empty slots, it ran one after the other.  The
  // slots stay where they are (the manager's lanes pass over the empty ones as ever); a view is a map from item to
  // (slot, level), and the iterate of the under-relaxation changes hands once, when the view is entered.  Same
  // operations, same operands: a view changes who works on an item, never what is done to it.
  const int W = W0;
  double *const slots = slots0, *const screc = screc0;
  int *const sirec = sirec0;

  for (int i = tid; i < NL; i += blockDim.x) {
    c_zm[i] = p.zm[i];
    c_hm[i] = p.hm[i];
    c_t0[i] = p.tri0[i];
    c_t1[i] = p.tri1[i];
    c_rdz[i] = rcp_refine(p.zm[i] - p.zm[i + 1]);
    c_dtohk[i] = p.dto / p.hm[i];
  }
  if (tid == 0) { c_misc[0] = rcp_refine(p.hm[1]); c_misc[1] = rcp_refine(p.vonk); }
  for (int i = tid; i < W0 * I_COUNT; i += blockDim.x) sirec0[i] = 0;   // every slot PS_EMPTY
  if (tid < 32) s_flags[tid] = tid == 3 ? nz : 0;
  // The manager is wave 0 (the item map below gives it items in the first trip only).  Measured and dropped:
  // electing the wave that sits on a given SIMD, so that the serial chains of all workgroups of a CU share one
  // SIMD (0.8-0.9x), or one SIMD per workgroup chosen from blockIdx (0.97x); raising its s_setprio (0.97x).
  const int mgr = 0;
  // the queue of the XCD this workgroup runs on (launches of several steps; M0)
  const int my_xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20 /* HW_REG_XCC_ID, bits 3:0 */) & 15;
  if (tid == 0 && p.nsteps_launch > 1) {   // claim the XCD's own queue (s_flags[7]: the queue M0 draws from)
    int q = p.xcc_queue[my_xcc];
    if (q >= 0) {
      const int prev = atomicCAS((int *)p.qowner + q, 0, my_xcc + 1);
      if (prev != 0 && prev != my_xcc + 1) q = -1;   // (another XCD has adopted it already: cannot happen while its own workgroups start with the rest)
    }
    s_flags[7] = q;
  }

  // ---- work items (slot, level) ------------------------------------------------
  // Item `it` of the first trip belongs to thread `it`; the later trips are dealt to the waves other than the
  // manager's only (it has its serial phases to run): item nthreads + t*(nthreads-64) + (tid-64) in trip t+1.
  const int nthreads = blockDim.x;
  const int nitems = W0 * L;
  const int nhelp = nthreads > 64 ? nthreads - 64 : 64, tid2 = nthreads > 64 ? tid - 64 : tid;
  const int nitems_lm = W0 * nzp1;
  const unsigned Wmagic = W0 > 1 ? 0xFFFFFFFFu / (unsigned)W0 + 1u : 0u;   // it / W == umulhi(it, Wmagic) for it < 2^16 (a view of one slot: W == 1, not used)
  // a view needs a wave for the manager and a lane for every item of its columns beside it
  const int kmax_view = nthreads > 64 ? min(min((nthreads - 64) / L, (int)S_MAPMAX), p.view_kmax > 0 ? p.view_kmax : (int)S_MAPMAX) : 0;
  const bool solo_ok = kmax_view >= 1;
  const bool solo_perm = W0 == 1 && solo_ok;   // a workgroup of one slot works in the view of that slot from the start (and refills it as ever)
  const bool solo_dyn = p.mode == MCKPP_MODE_STEP && solo_ok && !solo_perm && p.solo_limit > 0;   // ... others may go into one
  // This thread's item of the first trip, in slot-major (it0) and in level-major order (it0_lm): its own number in the
  // view of every slot; in a view of a few slots the index that the item loops' own arithmetic turns into the mapped slot
  // and the level - slot L + level - 1 and (level - 1) W + slot - so that the loops are the same code in either view.
  // trip_base: where the items of the later trips start (a view has one trip: past every item).
  bool sparse = false;
  int it0 = tid, it0_lm = tid, trip_base = nthreads;
  auto set_view = [&](int kv) {   // kv == 0: every slot; else the kv slots of s_flags[S_MAP ..]
    sparse = kv > 0;
    if (!sparse) { it0 = tid; it0_lm = tid; trip_base = nthreads; return; }
    trip_base = 0x20000000;
    const int i0 = tid - 64;
    it0 = 0x3fffffff; it0_lm = 0x3fffffff;
    if (i0 >= 0 && i0 < kv * L) {
      const int vs = (int)__umulhi((unsigned)i0, Lmagic);
      it0 = s_flags[S_MAP + vs] * L + (i0 - vs * L);
    }
    if (i0 >= 0 && i0 < kv * nzp1) {
      const int lk = i0 / kv;   // (once per view)
      it0_lm = lk * W0 + s_flags[S_MAP + (i0 - lk * kv)];
    }
  };
  if (solo_perm && tid == 0) { s_flags[S_KVIEW] = 1; s_flags[S_MAP] = 0; }
  __syncthreads();
  if (solo_perm) set_view(1);

  const double lambda = 0.5;
  const double epsln16 = 1.e-16, Ricr = 0.30, eps01 = 0.1, cekman = 0.7, cmonob = 1.0, epsln20 = 1.e-20;
  const bool do_ocnint = p.mode == MCKPP_MODE_STEP || p.mode == MCKPP_MODE_PASS;
  const bool flux_diag = p.mode == MCKPP_MODE_STEP || p.mode == MCKPP_MODE_INIT;   // wU, wX(1:nz) of ocnstep / initialize_ocean
  // the iterate of the under-relaxation between passes: four scratch rows per (workgroup, slot), element
  // index = level-1, each element read and rewritten in place by the one item that owns it.  The block is
  // reused by every column the slot serves, so it stays in L2.
  // the iterate of this thread's first item and of its second one (its items of later trips: scratch; none of the
  // geometries the launcher chooses has a third trip)
  double rU = 0.0, rV = 0.0, rT = 0.0, rS = 0.0, r2U = 0.0, r2V = 0.0, r2T = 0.0, r2S = 0.0;
  const int LS = ps_scratch_ld(nzp1);
  const auto scr0 = p.scratch + (size_t)blockIdx.x * (size_t)W0 * (size_t)(4 * LS);

// one level-parallel phase: every active (slot, level) item, strided by the workgroup's threads
// (not unrolled: the compiler peeled and unrolled L1's item loop - equation of state and all - into 2.5 copies, 14 KB
// of an 88 KB kernel and 26 more VGPRs, for nothing in the reference-order mode and -1.5 % in the two-ended one)
#define PS_ITEMS_PRAGMA _Pragma("clang loop unroll(disable)")
#define FOR_ITEMS                                                                         \
  PS_ITEMS_PRAGMA                                                                         \
  for (int it_ = it0, t_ = 0; it_ < nitems; it_ = tid2 >= 0 ? trip_base + t_ * nhelp + tid2 : nitems, ++t_) { \
    const int slot = (int)__umulhi((unsigned)it_, Lmagic);                                \
    const int k = it_ - slot * L + 1;                                                     \
    int *const si = sirec + slot * I_COUNT;                                               \
    if (!si[I_ACT]) continue;                                                             \
    double *const my = slots + slot * SS;                                                 \
    double *const sc = screc + slot * C_COUNT;                                            \
    const int col = si[I_COL];                                                            \
    const bool act = k <= nzp1, actz = k <= nz, virt1 = k == nzp1 + 1, virt2 = k == nzp1 + 2; \
    const bool is1 = k == 1, isnz = k == nz, isnzp1 = k == nzp1;                          \
    const int kr = act ? k : 1;                                                           \
    const size_t ro = (size_t)col * p.ld;                                                 \
    const auto xs_ = scr0 + (size_t)slot * (size_t)(4 * LS) + (kr - 1);   /* iterate U, V, T, S of this item */ \
    const int first_ = t_;   /* trip 0 / 1: the item's iterate stays in registers (rU.. / r2U..) */      \
    auto row = [&](int a) -> strided<ROWS> { return strided<ROWS>{my + a}; };             \
    (void)sc; (void)actz; (void)virt1; (void)virt2; (void)is1; (void)isnz; (void)isnzp1; (void)kr; (void)ro; (void)xs_; (void)first_; (void)row;
#define END_ITEMS }
// The same items in level-major order (item = (level-1)*W + slot, the equation-of-state items left out), for a
// phase whose cost grows with depth and that talks to the others through LDS only (L2: its reference-level
// loop runs over the layers above a tenth of the item's depth).  A wave then holds a few levels of every slot
// - its lanes loop equally long - and a thread's first item comes from the top of the columns, its later ones
// from the bottom, so every thread gets a deep and a shallow item.
#define FOR_ITEMS_BY_LEVEL                                                                \
  for (int j_ = it0_lm, t_ = 0; j_ < nitems_lm; j_ = tid2 >= 0 ? trip_base + t_ * nhelp + tid2 : nitems_lm, ++t_) { \
    const int it_ = t_ == 0 ? j_ : nitems_lm - 1 - (j_ - nthreads);                       \
    const int k = (W == 1 ? it_ : (int)__umulhi((unsigned)it_, Wmagic)) + 1;   /* the magic of W = 1 is 2^32 */ \
    const int slot = it_ - (k - 1) * W;                                                   \
    int *const si = sirec + slot * I_COUNT;                                               \
    if (!si[I_ACT]) continue;                                                             \
    double *const my = slots + slot * SS;                                                 \
    double *const sc = screc + slot * C_COUNT;                                            \
    const int col = si[I_COL];                                                            \
    const bool act = true, actz = k <= nz;                                                \
    const bool is1 = k == 1, isnz = k == nz, isnzp1 = k == nzp1;                          \
    const size_t ro = (size_t)col * p.ld;                                                 \
    auto row = [&](int a) -> strided<ROWS> { return strided<ROWS>{my + a}; };             \
    (void)sc; (void)act; (void)actz; (void)is1; (void)isnz; (void)isnzp1; (void)ro; (void)row;

// Level-major order again, plainly rising (item = (level-1)*W + slot, a thread's later items are deeper): for the
// phases that only have work down to some level - the waves that hold deeper levels fall through.
#define FOR_ITEMS_RISING                                                                  \
  for (int it_ = it0_lm, t_ = 0; it_ < nitems_lm; it_ = tid2 >= 0 ? trip_base + t_ * nhelp + tid2 : nitems_lm, ++t_) { \
    const int k = (W == 1 ? it_ : (int)__umulhi((unsigned)it_, Wmagic)) + 1;              \
    const int slot = it_ - (k - 1) * W;                                                   \
    int *const si = sirec + slot * I_COUNT;                                               \
    if (!si[I_ACT]) continue;                                                             \
    double *const my = slots + slot * SS;                                                 \
    double *const sc = screc + slot * C_COUNT;                                            \
    const bool actz = k <= nz, is1 = k == 1, isnz = k == nz;                              \
    auto row = [&](int a) -> strided<ROWS> { return strided<ROWS>{my + a}; };             \
    (void)sc; (void)actz; (void)is1; (void)isnz; (void)row;

  // =========================== manager phases (wave 0) ===========================
  // M0: slots whose column has finished pull the next one from the queue.
  // A launch may cover several model steps (p.nsteps_launch; mckpp_hip_step with nsteps > 1): the queue then holds
  // ncol x nsteps tickets in step-major order - ticket t is step t / ncol of column t mod ncol - and a column's step
  // may start once its previous step has been finished, by whichever workgroup had it: there is no barrier across
  // the chip between the steps.  One column that runs to itermax (200 passes where the others take 6: 4 ms more for
  // the whole launch, in one step of seven of a long run - profiles/r04/step_series.txt) then delays nothing but
  // its own next step.  p.done[c] counts the steps of the launch column c has completed: published after the
  // barrier behind the stores of the finishing step (every wave's vmcnt drained), read - with an agent-scope acquire,
  // which drops this CU's stale L1 lines - before the next step's loads.
  // A COLUMN STAYS ON ONE XCD for the whole launch: the per-XCD L2s are not coherent with each other (two steps of a
  // column on two XCDs would leave two dirty copies of its diagnostic rows, written back in any order; making every
  // step's 200 KB of stores visible across XCDs cost 10 % as an L2 write-back per finish round).  There is a queue
  // per XCD - column c belongs to queue c mod nq, a workgroup draws from the queue of the XCD it runs on, read from
  // the hardware (HW_REG_XCC_ID, mapped to 0..nq-1 by a probe at mckpp_hip_init) - so a column's rows live in one L2,
  // the coherence point of all CUs that ever touch them, and plain stores are enough.
  // A slot whose ticket is not ready keeps it (PS_WAIT) and asks again at every later M0.
  auto M0 = [&]() {
    int lane = lane_k; asm volatile("" : "+v"(lane));
    bool a = false, wt = false, sticky = false;
    const bool multi = p.nsteps_launch > 1;
    int *msi = sirec + (lane < W ? lane : 0) * I_COUNT;
    double *msc = screc + (lane < W ? lane : 0) * C_COUNT;
    int st = PS_DONE, c = 0, step = 0;
    bool want = false, ready = false, idle = false, dropped = false;
    // Who starts a column's next step.  A step is started by whoever first moves the column's count of started steps
    // (p.done[ncol + c]) from s to s+1, and there are two who try:
    //  - the slot that draws its ticket, if it finds the step before complete (p.done[c] >= s).  If it does not, it DROPS
    //    the ticket and draws the next - it does not hold it: a column on its way to itermax falls behind the queue by
    //    tens of steps in a long launch, and every ticket held for it was a slot that waited, empty (with fewer columns
    //    than slots the whole launch then moved in lockstep with its slowest column);
    //  - the slot that finishes the step before, if that ticket is out already (the queue's head is beyond it): the
    //    column goes on where it is, without a gap in the very chain the launch ends with.  A column whose next ticket
    //    has not been drawn ends here as ever - the queue's order is what keeps the workgroups' rounds together - and
    //    the ticket, when it is drawn, finds the step complete.
    // No step is lost between the two: the finishing slot publishes the step (an atomic exchange, waited for) BEFORE it
    // reads the head, the drawing slot has moved the head (an atomic add whose result names the column) BEFORE it reads
    // the column's count - if the one misses the other's write, the other sees the one's.  A column that goes on this
    // way counts as a straggler from its first pass (it is behind).
    bool cont = false;
    if (multi) {
      const bool pub = lane < W && msi[I_STATE] == PS_ACTIVE && msi[I_FIN] == F_FINAL;
      if (pub) {
        const int fc = msi[I_COL], fs = msi[I_STEP];
        int was = __hip_atomic_exchange(p.done + fc, fs + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(was) : : "memory");   // performed, before the head is read
        if (fs + 1 < p.nsteps_launch) {
          const int q = fc % p.nqueues, nloc = (p.ncol - q + p.nqueues - 1) / p.nqueues;
          const int tnext = (fs + 1) * nloc + fc / p.nqueues;   // c = q + nq j is ticket step nloc + j of queue q
          if (__hip_atomic_load((int *)p.qhead + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > tnext)
            cont = atomicCAS((int *)p.done + p.ncol + fc, fs + 1, fs + 2) == fs + 1;
          if (cont) { c = fc; step = fs + 1; want = true; }
        }
      }
    }
    const bool any_cont = __ballot(cont) != 0ull;
    // the workgroup is in a view of a few slots and some of their columns go on: nothing else starts here until they are
    // done - a ticket held for another column waits that long (its column's later steps wait for it anyway)
    const int kv = solo_perm ? 0 : s_flags[S_KVIEW];
    const bool solo_on = kv > 0 && (any_cont || __ballot(lane < W && sirec[lane * I_COUNT + I_STATE] == PS_ACTIVE && sirec[lane * I_COUNT + I_FIN] != F_FINAL) != 0ull);
    if (lane < W) {
      st = msi[I_STATE];
      if (st == PS_ACTIVE && msi[I_FIN] == F_FINAL) st = PS_EMPTY;   // its outputs are stored (barrier before M0)
      if (st == PS_WAIT && !solo_on) { c = msi[I_COL]; step = msi[I_STEP]; want = true; }
    }
    // Refills come in rounds.  A workgroup's slots that start together finish together - every column takes the
    // same six passes in the steady state - and each refill costs the whole workgroup a finish round, a pass that
    // forms every level's Richardson numbers (no guess for a new column) and a cold L1; a slot that finishes out of
    // step - its column needed seven passes, or two hundred - would from then on cause all that in a pass of its
    // own, for the rest of the launch (a launch of ten steps measured 7.5 ms per step at 100 levels where its steps
    // one by one took 6.0).  So a slot that comes free alone waits, empty, for the next round: slots are refilled
    // when at least half of those that take part in the rounds are free - a column past its twelfth pass does not
    // (it is on its way to itermax: nobody waits for it, and when it ends its slot waits for the others).
    const int n_empty = __popcll(__ballot(lane < W && st == PS_EMPTY && !cont));
    const int n_busy = __popcll(__ballot(lane < W && st == PS_ACTIVE && sirec[lane * I_COUNT + I_NPASS_TRY] <= 12));
    bool refill = 2 * n_empty >= n_empty + n_busy;
    // Stragglers.  A column past its solo_after-th pass of a try - or one that went to itermax in its previous step: they
    // do so step after step - is on its way to itermax: its chain of ~200 passes, each as long as a pass of the whole
    // workgroup, is what a long run waits for in the end (a launch of many steps: that column's steps follow each other;
    // a launch per step: the launch ends with it).  While the device holds few of them (p.sync[0]: their number, kept by
    // every workgroup's M0), a workgroup that has one does not refill its other slots: they run empty within a step, and
    // the pass - now of one column, in a view of its own (G_late) - takes 0.6 of the time.  The slots left idle are
    // a workgroup's share of the device per straggler; with many stragglers (the second step from an analytic start:
    // 14 % of the columns) nothing is left idle.
    bool drain = false;
    if (p.solo_limit > 0 && p.mode == MCKPP_MODE_STEP) {
      const int after = p.solo_after;
      const bool was = lane < W && msi[I_STRAG] != 0;
      const bool gone = was && st != PS_ACTIVE;                                             // finished (its outputs are stored)
      const bool is = lane < W && st == PS_ACTIVE && (was || msi[I_NPASS_TRY] > after);
      if (is && !was) msi[I_STRAG] = 1;
      if (gone) msi[I_STRAG] = 0;
      const int delta = __popcll(__ballot(is && !was)) - __popcll(__ballot(gone));
      int glob = 0;
      if (__ballot(is) != 0ull || delta != 0 || any_cont) {
        if (lane == 0) glob = delta != 0 ? atomicAdd((int *)p.sync, delta) + delta
                                         : __hip_atomic_load((int *)p.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        glob = __shfl(glob, 0);
      }
      drain = (__ballot(is) != 0ull || any_cont) && glob <= p.solo_limit;
      if (solo_on) drain = true;
      else if (kv > 0 && lane == 0) s_flags[S_KVIEW] = 0;   // its columns are done: back to the view of every slot
      if (drain) refill = false;
    }
    if (lane < W) {
      if (st == PS_EMPTY) { msi[I_FIN] = F_NONE; msi[I_ACT] = 0; }
      if (st == PS_EMPTY && refill && !cont) {
        if (!multi) {
          const int t = atomicAdd((int *)p.qhead, 1);
          if (t >= p.ncol) { st = PS_DONE; msi[I_ACT] = 0; }
          else { c = t; want = true; }
        } else {
          // the queue this workgroup draws from: its XCD's own, later one it has adopted (below), or none
          msi[I_ACT] = 0;
          const int q = s_flags[7];
          const int nloc = q < 0 ? 0 : (p.ncol - q + p.nqueues - 1) / p.nqueues;   // its columns: c = q + nq j
          for (;;) {   // (every trip takes a ticket off the queue)
            const int t = nloc > 0 ? atomicAdd((int *)p.qhead + q, 1) : 0;
            if (nloc <= 0 || t >= nloc * p.nsteps_launch) { idle = true; break; }   // stays PS_EMPTY: the workgroup looks for another queue
            step = t / nloc;
            c = q + p.nqueues * (t - step * nloc);
            int *const started = (int *)p.done + p.ncol + c;
            if (__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > step) continue;   // its column went on by itself
            if (step > 0 && __hip_atomic_load(p.done + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < step) continue;   // ... and will
            if (atomicCAS(started, step, step + 1) != step) continue;
            want = true; ready = true;
            break;
          }
        }
      }
      if (cont || ready) ready = true;
      else if (want && multi) {   // (a ticket held from an earlier M0: none is, since tickets are dropped - kept for the adoption path's sake)
        // a ticket: void if the step has been started already (its column went on by itself, above); ready once the
        // step before is complete, and then this slot's if it is the one to move the count of started steps
        int *const started = (int *)p.done + p.ncol + c;
        bool mine = false;
        if (__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= step) {
          ready = step == 0 || __hip_atomic_load(p.done + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= step;
          mine = !ready || atomicCAS(started, step, step + 1) == step;
        }
        if (!mine) { want = false; ready = false; dropped = true; st = PS_EMPTY; msi[I_ACT] = 0; }   // (M0 again at the next iteration: the slot draws another)
      } else if (want) ready = true;
    }
    if (multi && __ballot(want && ready && step > 0) != 0ull)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // before anything of those columns is read (here, and by every thread after the barrier)
    bool more = false;
    if (multi && __ballot(idle) != 0ull) {
      // This workgroup's queue is used up (or its XCD has none).  A queue nobody draws from - no workgroup of the
      // launch runs on its XCD: HIP promises nothing about placement - must still be served, and by ONE XCD: the
      // first to ask owns it (qowner: 0 free, else hardware XCC id + 1), and every workgroup of that XCD may then
      // draw from it.  Own queues are claimed at the start of the kernel, so only a queue without workgroups of its
      // own is ever adopted.
      int newq = -1;
      if (lane == 0) {
        for (int qq = 0; qq < p.nqueues && newq < 0; ++qq) {
          const int nl = (p.ncol - qq + p.nqueues - 1) / p.nqueues;
          if (nl <= 0 || __hip_atomic_load((int *)p.qhead + qq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nl * p.nsteps_launch) continue;
          const int prev = atomicCAS((int *)p.qowner + qq, 0, my_xcc + 1);
          if (prev == 0 || prev == my_xcc + 1) newq = qq;
        }
        s_flags[7] = newq;
      }
      more = __shfl(newq, 0) >= 0;   // the idle slots draw from it at the next M0 (which `waiting` asks for)
      if (idle && !more) st = PS_DONE;
    }
    if (lane < W) {
      if (want && !ready) {
        st = PS_WAIT;
        msi[I_ACT] = 0; msi[I_COL] = c; msi[I_STEP] = step;
      } else if (want) {
        {
          st = PS_ACTIVE;
          const auto ci = p.ci + (size_t)c * MCKPP_CI;
          const auto cs = p.cs + (size_t)c * MCKPP_CS;
          int old = ci[CI_OLD], newi = ci[CI_NEW], status = 0;
          if (old < 0 || old > 1) { old = newi; status |= 16; }
          if (newi < 0 || newi > 1) { newi = old; status |= 16; }
          msi[I_ACT] = 1; msi[I_COL] = c; msi[I_STEP] = step; msi[I_OLD] = old; msi[I_NEW] = newi; msi[I_JER] = ci[CI_JERLOV];
          msi[I_INITFLAG] = (p.mode == MCKPP_MODE_INIT) ? 1 : ci[CI_INITFLAG];   // initialize_ocean.F90:59
          msi[I_LOCEAN] = ci[CI_LOCEAN];
          msi[I_STATUS] = status; msi[I_NPASS] = 0; msi[I_NPASS_TRY] = 0; msi[I_ICONV] = 0; msi[I_COMP] = 1;
          msi[I_NRESET] = 0; msi[I_KMIXN] = 0; msi[I_KBL] = 0; msi[I_LOAD] = 1; msi[I_BAD] = 0;
          msi[I_MAYBE] = (p.mode != MCKPP_MODE_STEP) ? 1 : 0;
          msi[I_KBLC] = 0x7fffffff; msi[I_NVIOL] = 0; msi[I_NU] = 0; msi[I_NV] = 0; msi[I_NF] = 0; msi[I_PAR] = 0;
          msi[I_L1A] = 0; msi[I_MAYBE_NEXT] = msi[I_MAYBE]; msi[I_TINY] = 0;
          // (at itermax in its previous step - ci holds that step's pass count: a straggler from its first pass on)
          sticky = p.solo_limit > 0 && p.mode == MCKPP_MODE_STEP && (ci[CI_NPASS] > 50 || cont);
          msi[I_STRAG] = sticky ? 1 : 0;
          s_flags[3] = nz;   // no guess for a new column: L3 forms the bulk Richardson numbers of every level
          msc[C_F] = cs[CS_F]; msc[C_WXNT0] = 0.0; msc[C_HMIXE] = 0.0; msc[C_HMIXN] = 0.0;
          msc[C_SREF] = cs[CS_SREF]; msc[C_SSURF] = cs[CS_SSURF]; msc[C_OCDEPTH] = cs[CS_OCDEPTH];
          double s1 = cs[CS_SFLUX1], s2 = cs[CS_SFLUX2], s3 = cs[CS_SFLUX3], s4 = cs[CS_SFLUX4], s5 = cs[CS_SFLUX5], s6 = cs[CS_SFLUX6];
          if (multi && p.series && (p.ntime + step - 1) % p.ndtocn == 0 && ci[CI_LOCEAN]) {
            // the forced run in one launch: this step is a flux update (mckpp_ocean_model_3D.F90:44-48) - mckpp_fluxes'
            // assembly of sflux(1:6) (fluxes_mod.F90:55-77; k_fluxes of the launch-per-step path, same operations) from
            // the step's record, kept in the column's record for the steps until the next update and for the host
            const auto f8 = p.series + (size_t)((p.ntime + step - 1) / p.ndtocn - p.series_rec0) * 8 * (size_t)p.ncol + c;
            const size_t n = (size_t)p.ncol;
            double taux = f8[0];
            const double tauy = f8[n], swf = f8[2 * n], lwf = f8[3 * n], lhf = f8[4 * n], shf = f8[5 * n], rain = f8[6 * n], snow = f8[7 * n];
            if ((taux == 0.0) && (tauy == 0.0)) taux = 1.e-10;
            if (!p.l_rest) {
              s1 = taux; s2 = tauy; s3 = swf;
              s4 = lwf + lhf + shf - snow * p.flsn;
              s5 = 1e-10;
              s6 = rain + snow + (lhf / p.el);
            } else {
              s1 = 1.e-10; s2 = 0.00; s3 = 300.00; s4 = -300.00; s5 = 0.00; s6 = 0.00;
            }
            cs[CS_SFLUX1] = s1; cs[CS_SFLUX2] = s2; cs[CS_SFLUX3] = s3; cs[CS_SFLUX4] = s4; cs[CS_SFLUX5] = s5; cs[CS_SFLUX6] = s6;
          }
          msc[C_SFLUX1] = s1; msc[C_SFLUX2] = s2; msc[C_SFLUX3] = s3;
          msc[C_SFLUX4] = s4; msc[C_SFLUX5] = s5; msc[C_SFLUX6] = s6;
          {   // (M3 evaluates swfrac at -hbl in every pass: its constants from LDS, not through a memory round trip)
            const int jw = ci[CI_JERLOV];
            msc[C_JRFAC] = jer_rfac_c[jw]; msc[C_JA1] = jer_a1_c[jw]; msc[C_JA2] = jer_a2_c[jw];
            msc[C_JRA1] = jer_ra1_c[jw]; msc[C_JRA2] = jer_ra2_c[jw];
          }
        }
      }
      msi[I_STATE] = st;
      a = st == PS_ACTIVE;
      wt = st == PS_WAIT || (idle && more) || dropped;
    }
    const unsigned long long m = __ballot(a), mw = __ballot(wt);
    {
      const int nsticky = __popcll(__ballot(sticky));
      if (nsticky > 0 && lane == 0) atomicAdd((int *)p.sync, nsticky);
      // nothing will be refilled for now: stragglers are being left alone, or the queue has nothing more (G_late)
      const bool used_up = __ballot(lane < W && st == PS_DONE) != 0ull;
      if (lane == 0) s_flags[S_DRAIN] = (drain || used_up) ? 1 : 0;
      // a column has just been started here (a ticket that was waited for, a refill): a view the control asked for in the
      // pass before (G_early) no longer covers what the workgroup works on - it will ask again
      const bool started = __ballot(want && st == PS_ACTIVE) != 0ull;   // (every lane votes: not inside the lane-0 branch)
      if (lane == 0 && started) s_flags[S_GOVIEW] = 0;
    }
    if (lane == 0) { s_flags[0] = m != 0ull ? 1 : 0; s_flags[1] = 0; s_flags[6] = mw != 0ull ? 1 : 0; }
    if (multi) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the acquire's invalidate has completed before the barrier lets the other waves load
  };

  // M1: surface fluxes and friction velocity (verticalmixing_mod.F90:81-100), wXNT(0) (fluxes_mod.F90:110-116)
  auto M1 = [&]() {
    int lane = lane_k; asm volatile("" : "+v"(lane));
    if (lane < W) {
      int *msi = sirec + lane * I_COUNT;
      double *msc = screc + lane * C_COUNT;
      if (msi[I_ACT]) {
        msi[I_LOAD] = 0;
        msi[I_L1A] = 0;
        msi[I_PAR] = msi[I_PAR] ^ 1;   // L1 has just written the other copy of the iterate's level-1 temperature
        // Level 1's density, specific heat and expansion coefficients: evaluated here, under L2, from the T and S
        // its L1 item worked on (that item itself needs sigma-0 only on passes that cannot be the last)
        double talpha0, sbeta0, s0_1;
        abk80_dev(msc[X_S1], msc[X_T1], -c_zm[1], talpha0, sbeta0, s0_1);
        const double rho0 = 1000. + s0_1, cp0 = cpsw_dev(msc[X_S1], msc[X_T1], -c_zm[1]);
        const double rhoh2o = msc[X_RHOH2O], rhob = msc[X_RHOB];
        const double sflux1 = msc[C_SFLUX1], sflux2 = msc[C_SFLUX2], sflux3 = msc[C_SFLUX3], sflux4 = msc[C_SFLUX4],
                     sflux5 = msc[C_SFLUX5], sflux6 = msc[C_SFLUX6];
        const double Ssurf = msc[C_SSURF];
        const double r_rho0 = rcp_refine(rho0), rho0cp0 = rho0 * cp0, r_rc = rcp_refine(rho0cp0);
        const double wU0_1 = div_fast(-sflux1, rho0, r_rho0);
        const double wU0_2 = div_fast(-sflux2, rho0, r_rho0);
        const double tau = __builtin_sqrt(sflux1 * sflux1 + sflux2 * sflux2) + 1.e-16;
        const double ustar = __builtin_sqrt(div_fast(tau, rho0, r_rho0));
        const double wX0_1 = div_fast(div_fast(-sflux4, rho0, r_rho0), cp0, rcp_refine(cp0));
        const double wX0_2 = div_fast(Ssurf * sflux6, rhoh2o, rcp_refine(rhoh2o)) +
                             div_fast((Ssurf - p.sice) * sflux5, rhob, rcp_refine(rhob));
        const double B0 = -p.grav * (talpha0 * wX0_1 - sbeta0 * wX0_2);
        const double B0sol = div_fast(p.grav * talpha0 * sflux3, rho0cp0, r_rc);
        const wscale_u wu = wscale_prepare(ustar);
        const double fa = __builtin_fabs(msc[C_F]) + epsln16;
        msc[C_B0] = B0; msc[C_B0SOL] = B0sol; msc[C_USTAR] = ustar; msc[C_UFRAC] = wu.ufrac; msc[C_UCUBE] = wu.ucube;
        msi[I_JU] = wu.ju;
        msc[C_HEK] = div_fast(cekman * ustar, fa, rcp_refine(fa));   // Ekman depth scale, bldepth_mod.F90:158
        msc[C_WU01] = wU0_1; msc[C_WU02] = wU0_2; msc[C_WX01] = wX0_1; msc[C_WX02] = wX0_2;
        msc[C_RHO0CP0] = rho0cp0; msc[C_RRC] = r_rc;
        if (ntime >= 1) msc[C_WXNT0] = div_fast(-sflux3 * p.swdk_tab[msi[I_JER] * p.ldc], rho0cp0, r_rc);
      }
    }
  };

  // M3: boundary-layer depth from the first level with hmin < -zm(k) (bldepth_mod.F90:161-201) and the
  //     slot-uniform part of blmix (blmix_mod.F90:62-100, 136-149).
  // One lane per (slot, species): momentum and temperature (= salinity without double diffusion: the same numbers), with
  // double diffusion all three.  The phase is 600 instructions at six cycles apiece on one wave whatever the number of
  // lanes, and most of it comes in pairs - swfrac's two exponentials, the two wscale look-ups (sigma = 1 | epsilon and the
  // grid level above kbl), and per species the gradient of the interior diffusivity at kn, gat1, dat1, dkm1 - so the
  // lanes of a slot each take one of a pair (the same instructions on their own operands), exchange the three values
  // the others need through the crossbar, and store their own species' results; what is common to the species every
  // lane forms for itself.  Same operations on the same operands as one lane per slot.
  auto M3 = [&]() {
    int lane = lane_k; asm volatile("" : "+v"(lane));
    constexpr int NSP = DD ? 3 : 2;
    const int sl = NSP == 2 ? lane >> 1 : (int)__umulhi((unsigned)lane, 0x55555556u), j = lane - sl * NSP;
    if (lane < NSP * W) {
      int *msi = sirec + sl * I_COUNT;
      double *msc = screc + sl * C_COUNT;
      if (msi[I_ACT]) {
        double *mrow = slots + sl * SS;
        const int kc = msi[I_KBLC];
        if (j == 0) msi[I_KBLC] = 0x7fffffff;
        int kbl = nz;
        double hbl = -c_zm[nz];
        if (kc <= nz) { kbl = kc; hbl = mrow[kc * ROWS + Q_YS]; }
        const double B0 = msc[C_B0], B0sol = msc[C_B0SOL], ustar = msc[C_USTAR];
        wscale_u wu;
        wu.ju = msi[I_JU]; wu.ufrac = msc[C_UFRAC]; wu.ustar = ustar; wu.ucube = msc[C_UCUBE];
        double bfsfc;
        {   // swfrac_dev(-1.0, hbl, jer) with the slot's copies of the Jerlov constants (same operations): this lane's
            // exponential of the two
          const double rmin = -80.;
          const double zf = hbl * -1.0;
          const double aj = msc[j == 0 ? (int)C_JA1 : (int)C_JA2], raj = msc[j == 0 ? (int)C_JRA1 : (int)C_JRA2];
          const double ej = mckpp_exp(dmax2(div_fast(zf, aj, raj), rmin));
          const double e1 = __shfl(ej, lane - j), e2 = __shfl(ej, lane - j + 1);
          const double rfac = msc[C_JRFAC];
          bfsfc = rfac * e1 + (1. - rfac) * e2;
        }
        bfsfc = B0 + B0sol * (1. - bfsfc);
        const double stable = 0.5 + dsign(0.5, bfsfc);
        bfsfc = bfsfc + stable * epsln16;
        const double caseA = 0.5 + dsign(0.5, -c_zm[kbl] - 0.5 * c_hm[kbl] - hbl);
        const double r_hbl = rcp_refine(hbl);
        // the two look-ups of this phase (blmix_mod.F90:64-66 at sigma = 1 | epsilon, :136-141 at the grid level
        // above kbl): lane 0 of the slot the first, the others the second; momentum wants wm of both, the scalars ws
        const double sig_k = div_fast(-c_zm[kbl - 1], hbl, r_hbl);
        const double sg_a = j == 0 ? 1.0 : sig_k, sg_b = j == 0 ? eps01 : dmin2(sig_k, eps01);
        double wm_j, ws_j;
        wscale_finish(p, wu, wscale_fetch(p, wu, stable * sg_a + (1. - stable) * sg_b, hbl, bfsfc), wm_j, ws_j);
        const double got = __shfl(j == 0 ? ws_j : wm_j, j == 0 ? lane + 1 : lane - j);
        const double w_1 = j == 0 ? wm_j : got;   // this species' velocity scale at sigma = 1 | epsilon ...
        const double w_k = j == 0 ? got : ws_j;   // ... and at the grid level above kbl
        // this lane's species: 0 momentum, 1 salinity, 2 temperature (without double diffusion the interior difs and dift
        // are the same numbers, and so is everything formed from them here: the temperature lane stores both)
        const int m = j == 0 ? 0 : DD ? j : 2;
        double gat1, dat1;
        {
          int ifx = (int)(caseA + epsln20);
          int kn = ifx * (kbl - 1) + (1 - ifx) * kbl;
          double hmkn = c_hm[kn], hmkn1 = c_hm[kn + 1];
          const double r_hmkn = rcp_refine(hmkn), r_hmkn1 = rcp_refine(hmkn1);
          double delhat = 0.5 * hmkn - c_zm[kn] - hbl;
          double R = 1.0 - div_fast(delhat, hmkn, r_hmkn);
          const strided<ROWS> dd{mrow + (m == 0 ? (int)Q_DM : m == 1 ? (int)Q_DS : (int)Q_DT)};
          const double dvdzup = div_fast(dd[kn - 1] - dd[kn], hmkn, r_hmkn);
          const double dvdzdn = div_fast(dd[kn] - dd[kn + 1], hmkn1, r_hmkn1);
          const double dp = 0.5 * ((1. - R) * (dvdzup + __builtin_fabs(dvdzup)) + R * (dvdzdn + __builtin_fabs(dvdzdn)));
          const double dh = dd[kn] + dp * delhat;
          double u4 = ((ustar * ustar) * ustar) * ustar;
          const double u4e = u4 + epsln20;
          double f1 = div_fast(stable * 5.0 * bfsfc, u4e, rcp_refine(u4e));
          const double we = w_1 + epsln20, r_we = rcp_refine(we);
          gat1 = div_fast(div_fast(dh, hbl, r_hbl), we, r_we);
          dat1 = div_fast(-dp, we, r_we) + f1 * dh;
          dat1 = dmin2(dat1, 0.);
        }
        double dkm1;
        {
          const double sig = sig_k;
          double a1 = sig - 2.;
          double a2 = 3. - 2. * sig;
          double a3 = sig - 1.;
          double G = a1 + a2 * gat1 + a3 * dat1;
          dkm1 = hbl * w_k * sig * (1. + sig * G);
        }
        msc[C_GAT1 + m] = gat1; msc[C_DAT1 + m] = dat1; msc[C_DKM1 + m] = dkm1;
        if (!DD && j == 1) { msc[C_GAT1 + 1] = gat1; msc[C_DAT1 + 1] = dat1; msc[C_DKM1 + 1] = dkm1; }
        if (j == 0) {
          msi[I_KBL] = kbl;
          msc[C_HBL] = hbl;
          msc[C_RHBL] = r_hbl; msc[C_STABLE] = stable; msc[C_BFSFC] = bfsfc; msc[C_CASEA] = caseA;
        }
      }
    }
  };

  // G: ocnstep control of a pass (ocnstep_mod.F90:122-192), one lane per slot.  Whether a column goes on
  // iterating depends on this pass's boundary-layer depth only, so the decision is taken well before the sweeps
  // (G_early: behind the manager's L6 items, in what would be its wait at that barrier) - a slot that goes on can
  // then start the next pass's L1 while its V sweep still runs - and what
  // the sweeps and the rest of this pass still need (zero-pivot flag, this pass's `maybe last` flag) is settled
  // after them (G_late).
  // (not with double diffusion: its L1 stages neighbour values in rows the V sweep is reading)
  const bool l1_ahead = p.mode == MCKPP_MODE_STEP && nthreads > 64 && !(EXT && p.LDD);
  auto G_early = [&]() {
    int lane = lane_k; asm volatile("" : "+v"(lane));
    bool f_any = false;
    if (lane < W) {
      int *msi = sirec + lane * I_COUNT;
      double *msc = screc + lane * C_COUNT;
      if (msi[I_ACT]) {
        int fin = F_NONE;
        int status = msi[I_STATUS], npass_try = msi[I_NPASS_TRY], iconv = msi[I_ICONV];
        msi[I_NPASS] = msi[I_NPASS] + 1;
        if (p.mode != MCKPP_MODE_STEP) {
          fin = F_FINAL;
          msi[I_MAYBE_NEXT] = msi[I_MAYBE];
        } else {
          ++npass_try;
          const double hbl = msc[C_HBL];
          if (npass_try <= 3) {   // compulsory passes
            msc[C_HMIXE] = hbl;
          } else {
            const double hmixn = hbl, hmixe = msc[C_HMIXE];
            const int kmixn = msi[I_KBL];
            msc[C_HMIXN] = hmixn;
            msi[I_KMIXN] = kmixn;
            double tol = p.hmixtolfrac * c_hm[kmixn];
            if (kmixn == nzp1) tol = p.hmixtolfrac * c_hm[nz];
            if (__builtin_fabs(hmixn - hmixe) > tol) iconv = 0;
            else iconv = iconv + 1;
            bool go_on = false;
            if (iconv < 3) {
              if (npass_try < p.itermax) { msc[C_HMIXE] = hmixn; go_on = true; }
              else if (hmixn > hmixe) { msc[C_HMIXE] = hmixn; go_on = true; }
            }
            if (!go_on) {
              if (npass_try > (p.itermax + 1)) status |= 2;
              fin = F_TRAP;
              msi[I_NVIOL] = 0;
            }
          }
          msi[I_MAYBE_NEXT] = (npass_try >= 3 && (iconv >= 2 || npass_try + 1 >= p.itermax)) ? 1 : 0;
        }
        msi[I_STATUS] = status; msi[I_NPASS_TRY] = npass_try; msi[I_ICONV] = iconv;
        msi[I_FIN] = fin;
        msi[I_L1A] = (l1_ahead && fin == F_NONE) ? 1 : 0;
        f_any = fin != F_NONE;
      }
    }
    const unsigned long long m = __ballot(f_any);
    if (lane == 0) s_flags[1] = m != 0ull ? 1 : 0;
    // A few columns left in a workgroup that does not refill for now, every one of them on its way to itermax (or at
    // itermax in its previous step): from the next pass on the workgroup works in a view of their slots (the iterate
    // changes hands after that pass's L1, below).  (A slot that holds a ticket it waits to start keeps it until these
    // columns are done, M0: mostly it is one of these very columns' next step.)  Decided here, in the manager's wait
    // behind its L6 items.
#ifdef MCKPP_PS_STAMPS
    {   // census of the stragglers' passes (p.sync[1..6]; MCKPP_LIST_DEBUG=1 prints it): theirs in a view | beside other
        // columns, the workgroup passes of either kind, and the active slots of those
      const bool sg = lane < W && sirec[lane * I_COUNT + I_STATE] == PS_ACTIVE && sirec[lane * I_COUNT + I_ACT] && sirec[lane * I_COUNT + I_STRAG];
      const int n = __popcll(__ballot(sg));
      const int nact = __popcll(__ballot(lane < W && sirec[lane * I_COUNT + I_STATE] == PS_ACTIVE && sirec[lane * I_COUNT + I_ACT]));
      if (lane == 0 && n > 0) { atomicAdd((int *)p.sync + (sparse ? 1 : 2), n); atomicAdd((int *)p.sync + (sparse ? 3 : 4), 1); atomicAdd((int *)p.sync + (sparse ? 5 : 6), nact); }
    }
#endif
    if (solo_dyn && !sparse) {
      const int f_kv = s_flags[S_KVIEW], f_drain = s_flags[S_DRAIN];
      if (f_kv == 0 && f_drain) {
        const int *msi = sirec + (lane < W ? lane : 0) * I_COUNT;
        const bool on = lane < W && msi[I_STATE] == PS_ACTIVE && msi[I_ACT];
        const bool ok = on && msi[I_FIN] == F_NONE && (msi[I_STRAG] || msi[I_NPASS_TRY] > p.solo_after);
        const unsigned long long m_on = __ballot(on), m_ok = __ballot(ok);
        const int kn = __popcll(m_on);
        if (kn >= 1 && kn <= kmax_view && m_ok == m_on) {
          if (on) s_flags[S_MAP + __popcll(m_on & ((1ull << lane) - 1ull))] = lane;
          if (lane == 0) s_flags[S_GOVIEW] = kn;
        }
      }
    }
  };
  auto G_late = [&]() {
    int lane = lane_k; asm volatile("" : "+v"(lane));
    if (lane < W) {
      int *msi = sirec + lane * I_COUNT;
      if (msi[I_ACT]) {
        if (p.mode != MCKPP_MODE_INIT && msi[I_BAD]) msi[I_STATUS] = msi[I_STATUS] | 1;
        msi[I_BAD] = 0;
        msi[I_TINY] = 0;
        msi[I_MAYBE] = msi[I_MAYBE_NEXT];
      }
    }
  };

  // ---- optional terms of the T and S right-hand sides (ocnint_mod.F90:97-215), level k of item (my, si, col):
  // relaxation / flux corrections / prescribed advection (rhsmod, solvers.F90:176-335, salinity only)
  auto ext_rhs = [&](double *my, const int *si, int col, int k, int kmixe, double To_k, double So_k, double &rhsT,
                     double &rhsS) {
    const double dto = p.dto;
    const auto xs = p.xs + (size_t)col * MCKPP_XS;
    const double rhok = my[k * ROWS + Q_RHO], cpk = my[k * ROWS + Q_CP];
    const size_t oin = (size_t)col * p.ld + (k - 1);
    if (k == 1) {
      if (p.L_RELAX_SST && !p.L_FCORR_WITHZ && !p.L_FCORR) {   // :97-114
        const double relax_sst = xs[XS_RELAX_SST], SST0 = xs[XS_SST0];
        double fc = 0.0;
        if (relax_sst > 1.e-10) {
          if (!p.L_RELAX_CALCONLY) rhsT = rhsT + dto * relax_sst * (SST0 - To_k) * p.dm[kmixe] / c_hm[1];
          fc = relax_sst * (SST0 - To_k) * p.dm[kmixe] * rhok * cpk;
        }
        p.cs[(size_t)col * MCKPP_CS + CS_FCORR] = fc;
      }
      if (p.L_FCORR && !p.L_RELAX_SST && !p.L_FCORR_WITHZ)     // :121-125
        rhsT = rhsT + dto * xs[XS_FCORR_TWOD] / (rhok * cpk * c_hm[1]);
    }
    double tinc = 0.;                                           // :133-160
    if (p.L_FCORR_WITHZ && !p.L_FCORR) tinc = dto * p.fcorr_withz[oin] / (rhok * cpk);
    if (p.L_RELAX_OCNT) tinc = tinc + dto * xs[XS_RELAX_OCNT] * (p.ocnT_clim[oin] - To_k);
    rhsT = rhsT + tinc;
    const double ocnTcorr = tinc * rhok * cpk / dto;
    // prescribed advection of salinity, rhsmod with jsclr = 2 (:179-184)
    const auto ai = p.adv_i + (size_t)col * (p.maxmodeadv + 1);
    const auto ad = p.adv_d + (size_t)col * (p.maxmodeadv + 1);
    const int nmode = ai[0];
    const int nzi = nz, km = kmixe;
    for (int im = 0; im < nmode; ++im) {
      const int mode = ai[1 + im];
      if (mode <= 0) continue;
      const double fact = dto * ad[im] * 0.033;
      if (mode == 1) {
        if (k == 1) rhsS = rhsS + fact / c_hm[1];
      } else if (mode == 2) {
        const double delta = p.hsum[km - 1];
        if (k <= km - 1) rhsS = rhsS + fact / delta;
      } else if (mode == 3) {
        const double delta = p.hsum[nzi];
        if (k <= nzi) rhsS = rhsS + fact / delta;
      } else if (mode == 4) {
        const int nzend = nzi - 1;
        int n1 = 0;
        do { n1 = n1 + 1; } while (c_zm[n1] >= -100. && n1 < nzp1);
        double delta = 0.0;
        for (int n = n1; n <= nzend; ++n) delta = delta + c_hm[n];
        if (k >= n1 && k <= nzend) rhsS = rhsS + fact / delta;
      } else if (mode == 5) {
        if (k == nzi) rhsS = rhsS + fact / c_hm[nzi];
      } else if (mode == 6 || mode == 7) {
        int n1, n2 = 0;
        double depth, dmax, delta = 0.0;
        if (mode == 6) { n1 = 1; depth = c_hm[1]; dmax = p.dm[km] - 0.5 * (c_hm[km] + c_hm[km - 1]); }
        else { n1 = km - 1; depth = p.dm[km] - 0.5 * c_hm[km]; dmax = 100.; }
        for (int n = n1; n <= nzi; ++n) {
          n2 = n;
          delta = delta + c_hm[n];
          depth = depth + c_hm[n + 1];
          if (depth >= dmax) break;
        }
        if (k >= n1 && k <= n2) rhsS = rhsS + fact / delta;
      }
    }
    double sinc = 0.;                                           // :187-213
    if (p.L_SFCORR_WITHZ && !p.L_SFCORR) sinc = dto * p.sfcorr_withz[oin];
    if (p.L_RELAX_SAL) sinc = sinc + dto * xs[XS_RELAX_SAL] * (p.sal_clim[oin] - So_k);
    rhsS = rhsS + sinc;
    // tinc_fcorr of the latest pass is what check_profile adds to (overrides.F90:87-88): always stored
    const size_t o = (size_t)col * p.ld + k;
    p.tinc_fcorr[o] = tinc;
    if (si[I_MAYBE]) { p.ocnTcorr[o] = ocnTcorr; p.sinc_fcorr[o] = sinc; p.scorr[o] = sinc / dto; }
  };

  // ---- L1 of one item: under-relaxation of the iterate against the last solution (ocnstep_mod.F90:123-132 /
  // :142-151; for a new or retried column the extrapolation of :91-112), equation of state.  `part`: everything;
  // everything but V (run by the item's own thread while the manager wave sweeps V: nothing here but the
  // relaxation of V depends on that sweep); V only (what is then left for the L1 phase).
  enum { L1_FULL = 0, L1_ALL_BUT_V = 1, L1_V_ONLY = 2 };
  auto L1_item = [&](const int k, const int kr, int *const si, double *const my, double *const sc, const size_t ro,
                     const bool act, const bool virt1, const bool virt2, const bool is1, const auto xs_,
                     const int first_, const int part) {
    auto row = [&](int a) -> strided<ROWS> { return strided<ROWS>{my + a}; };
    if (part == L1_V_ONLY) {
      if (act) {
        double V = first_ == 0 ? rV : first_ == 1 ? r2V : xs_[LS];
        V = lambda * V + (1 - lambda) * row(Q_YV)[k];
        if (first_ == 0) rV = V; else if (first_ == 1) r2V = V; else xs_[LS] = V;
        row(Q_YV)[k] = V;
      }
      return;
    }
    const bool with_v = part == L1_FULL;
    const int maybe = part == L1_ALL_BUT_V ? si[I_MAYBE_NEXT] : si[I_MAYBE];
    const int ldf = with_v ? si[I_LOAD] : 0, par = si[I_PAR];
    const size_t o = ro + (kr - 1);
    double U = 0, V = 0, T = 0, S = 0;
    if (p.mode == MCKPP_MODE_STEP) {
      double yu, yv, yt, ys;
      if (ldf != 0) {   // the relaxation memory equals the new iterate (Ux = U, ocnstep_mod.F90:105,110)
        const int old = si[I_OLD], newi = si[I_NEW];
        const double uo = act ? p.Us[old][o] : 0.0, un = act ? p.Us[newi][o] : 0.0;
        const double vo = act ? p.Vs[old][o] : 0.0, vn = act ? p.Vs[newi][o] : 0.0;
        const double to = p.Ts[old][o], tn = p.Ts[newi][o];
        const double so = act ? p.Ss[old][o] : 0.0, sn = act ? p.Ss[newi][o] : 0.0;
        U = 2. * un - uo;
        V = 2. * vn - vo;
        T = 2. * tn - to;
        S = 2. * sn - so;
        yu = U; yv = V; yt = T; ys = S;
      } else {
        if (first_ == 0) { U = rU; T = rT; S = rS; if (with_v) V = rV; }
        else if (first_ == 1) { U = r2U; T = r2T; S = r2S; if (with_v) V = r2V; }
        else if (act) { U = xs_[0]; T = xs_[2 * LS]; S = xs_[3 * LS]; if (with_v) V = xs_[LS]; }
        if (!act) T = sc[C_T1X + par];   // the two EOS items follow level 1, whose item rewrites it now
        yu = row(Q_YU)[kr]; yv = with_v ? row(Q_YV)[kr] : 0.0; yt = row(Q_YT)[kr]; ys = row(Q_YS)[kr];
      }
      // under-relaxation, ocnstep_mod.F90:123-132 / :142-151 (an EOS item follows level 1's temperature)
      T = lambda * T + (1 - lambda) * yt;
      if (act) {
        U = lambda * U + (1 - lambda) * yu;
        if (with_v) V = lambda * V + (1 - lambda) * yv;
        S = lambda * S + (1 - lambda) * ys;
        if (first_ == 0) { rU = U; rT = T; rS = S; if (with_v) rV = V; }
        else if (first_ == 1) { r2U = U; r2T = T; r2S = S; if (with_v) r2V = V; }
        else { xs_[0] = U; xs_[2 * LS] = T; xs_[3 * LS] = S; if (with_v) xs_[LS] = V; }
        if (is1) sc[C_T1X + (par ^ 1)] = T;
      }
    } else {
      U = act ? p.U[o] : 0.0; V = act ? p.V[o] : 0.0; T = p.T[o]; S = act ? p.S[o] : 0.0;
    }
    const double Sref = sc[C_SREF];
    const double zm1 = c_zm[1];
    const double zmk = c_zm[kr];
    double Sin = S + Sref, Pin = -zmk;
    const double Tin = T;
    if (virt1) { Sin = 0.0; Pin = -zm1; }
    if (virt2) { Sin = p.sice; Pin = -zm1; }
    // Only sigma-0 (density, buoyancy) feeds the pass; alpha, beta and cp of the levels are diagnostics of the
    // last vmix (and inputs of the optional physics), level 1's are formed in M1: passes that cannot be the
    // last evaluate a tenth of the equation of state.
    // (Optional physics: rho cp enters the right-hand sides under three switches, double diffusion needs alpha
    // and beta, and the correction diagnostics of a pass that may be the last need rho cp too.)
    const bool full_eos = (maybe && (p.diag || EXT)) ||
                          (EXT && (p.LDD || p.L_RELAX_SST || p.L_FCORR || p.L_FCORR_WITHZ));
    double s0, talpha = 0.0, sbeta = 0.0, cp = 0.0;
    if (full_eos) {
      abk80_dev(Sin, Tin, Pin, talpha, sbeta, s0);
      cp = cpsw_dev(Sin, Tin, Pin);
    } else {
      s0 = sig0_dev(Sin, Tin);
    }
    const double rho = 1000. + s0;
    const double buoy = div_fast(-p.grav * s0, 1000., 1. / 1000.);
    if (is1) { sc[X_T1] = Tin; sc[X_S1] = Sin; }
    if (virt1) sc[X_RHOH2O] = rho;
    if (virt2) sc[X_RHOB] = rho;
    if (act) { row(Q_YU)[k] = U; row(Q_YS)[k] = buoy; if (with_v) row(Q_YV)[k] = V; }
    if (p.diag && maybe) {   // what the last vmix leaves behind (types_transfer.F90:199-327)
      const size_t od = ro + k;
      if (act) { p.rho[od] = rho; p.cp[od] = cp; p.buoy[od] = buoy; p.talpha[od] = talpha; p.sbeta[od] = sbeta; }
      if (is1) { p.rho[od - 1] = rho; p.cp[od - 1] = cp; p.talpha[od - 1] = talpha; p.sbeta[od - 1] = sbeta; }
    }
    if constexpr (EXT) {
      if (act) { row(Q_RHO)[k] = rho; row(Q_CP)[k] = cp; }
      if (DD && p.LDD && act) {   // neighbours for alphaDT, betaDS
        row(Q_DM)[k] = talpha; row(Q_S1)[k] = sbeta; row(Q_S2)[k] = S; row(Q_BET)[k] = T;
      }
    }
  };

  // ---- L2 of one item (level k <= nzp1 of a slot): surface-layer reference averages, Ri pieces
  // (verticalmixing_mod.F90:111-137).  The reference averages (uref, vref, bref: the loop over the layers above a
  // tenth of the level's depth) feed Ritop and dVsq, which only bldepth's bulk Richardson number uses: they are
  // formed for the levels L3 will form that number for (`sums`), and later for the others should the scan of M2
  // ask for them - then from copies of the iterate's U and V in other rows (rowU, rowV), their own having been reused.
  auto ref_sums = [&](const int k, int *const si, double *const my, const int rowU, const int rowV, const bool allow_pre,
                      const bool actz, double &ur, double &vr, double &br) {
    auto row = [&](int a) -> strided<ROWS> { return strided<ROWS>{my + a}; };
    const strided<ROWS> aU = row(rowU), aV = row(rowV), aB = row(Q_YS);
    const double zmk = c_zm[k];
    const double zm1 = c_zm[1];
    const double U1 = aU[1], V1 = aV[1], Bu1 = aB[1];
    const strided<ROWS> aNU = row(Q_DM), aNV = row(Q_BET);   // whole-layer terms of U and V (L2a)
    const bool pre = allow_pre && p.l2pre != 0;
    const bool guard = si[I_TINY] != 0;
    const double zref = eps01 * zmk, rzref = rcp_refine(zref);
    double wz = dmax2(zm1, zref);
    ur = div_fast_guarded(U1 * wz, zref, rzref); vr = div_fast_guarded(V1 * wz, zref, rzref);
    br = div_fast(Bu1 * wz, zref, rzref);
    // The layers above zref (verticalmixing_mod.F90:118-131: wz = MIN(dz, zm(kl)-zref), del = 0.5 wz/dz).  Every
    // layer but the one zref lies in is taken whole - wz = dz, del = 0.5 exactly - and needs neither the minimum
    // nor the division; the one partial layer is the last of its lane and is done after the loop.
    bool live = actz;
    int klp = 0;
    double zk = zm1, Uk = U1, Vk = V1, Bk = Bu1;
    if (pre) {
      // the numerators of the whole layers, dz (U(kl) + 0.5 (U(kl+1) - U(kl))) and the same of V, do not depend on
      // the level whose reference values are being formed: L2a has put them into two rows, and has said whether
      // any of them is a tiny non-zero number (div_fast must not see those)
      for (int kl = 1; kl <= nz; ++kl) {
        live = live && !(zref >= zk);
        if (!__any(live)) break;
        const double zk1 = c_zm[kl + 1], nu = aNU[kl], nv = aNV[kl], Bk1 = aB[kl + 1];
        if (live) {
          const double dzk = zk - zk1;
          if (dzk > zk - zref) {   // the minimum is zm(kl)-zref < dz: partial layer (then zref > zm(kl+1): the last)
            klp = kl;
            live = false;
          } else {
            if (__builtin_expect(guard, 0)) {
              ur = ur - div_fast_guarded(nu, zref, rzref);
              vr = vr - div_fast_guarded(nv, zref, rzref);
            } else {
              ur = ur - div_fast(nu, zref, rzref);
              vr = vr - div_fast(nv, zref, rzref);
            }
            br = br - div_fast(dzk * (Bk + 0.5 * (Bk1 - Bk)), zref, rzref);
          }
        }
        zk = zk1; Bk = Bk1;
      }
    } else {
      for (int kl = 1; kl <= nz; ++kl) {
        live = live && !(zref >= zk);
        if (!__any(live)) break;
        const double zk1 = c_zm[kl + 1], Uk1 = aU[kl + 1], Vk1 = aV[kl + 1], Bk1 = aB[kl + 1];
        if (live) {
          const double dzk = zk - zk1;
          if (dzk > zk - zref) {
            klp = kl;
            live = false;
          } else {
            ur = ur - div_fast_guarded(dzk * (Uk + 0.5 * (Uk1 - Uk)), zref, rzref);
            vr = vr - div_fast_guarded(dzk * (Vk + 0.5 * (Vk1 - Vk)), zref, rzref);
            br = br - div_fast(dzk * (Bk + 0.5 * (Bk1 - Bk)), zref, rzref);
          }
        }
        zk = zk1; Uk = Uk1; Vk = Vk1; Bk = Bk1;
      }
    }
    if (klp) {
      const double zl = c_zm[klp], zl1 = c_zm[klp + 1];
      const double Ul = aU[klp], Ul1 = aU[klp + 1], Vl = aV[klp], Vl1 = aV[klp + 1], Bl = aB[klp], Bl1 = aB[klp + 1];
      const double dzk = zl - zl1, wz2 = zl - zref;
      const double del = div_fast(0.5 * wz2, dzk, c_rdz[klp]);
      ur = ur - div_fast_guarded(wz2 * (Ul + del * (Ul1 - Ul)), zref, rzref);
      vr = vr - div_fast_guarded(wz2 * (Vl + del * (Vl1 - Vl)), zref, rzref);
      br = br - div_fast(wz2 * (Bl + del * (Bl1 - Bl)), zref, rzref);
    }
  };
  auto L2_item = [&](const int k, int *const si, double *const my, double *const sc, const size_t ro, const bool actz,
                     const bool is1, const bool isnz, const bool isnzp1, const bool sums) {
    auto row = [&](int a) -> strided<ROWS> { return strided<ROWS>{my + a}; };
    const strided<ROWS> aU = row(Q_YU), aV = row(Q_YV), aB = row(Q_YS);
    const double U = aU[k], V = aV[k], buoy = aB[k];
    const double zmk = c_zm[k];
    if (sums) {
      double ur, vr, br;
      ref_sums(k, si, my, Q_YU, Q_YV, true, actz, ur, vr, br);
      const double zref = eps01 * zmk;
      const double Ritop = (zref - zmk) * (br - buoy);
      const double dVsq = (ur - U) * (ur - U) + (vr - V) * (vr - V);
      if (p.mode != MCKPP_MODE_STEP && isnz) { sc[C_UREFNZ] = ur; sc[C_VREFNZ] = vr; }
      if (actz) { row(Q_DT)[k] = Ritop; row(Q_DS)[k] = dVsq; }
    }
    if constexpr (EXT) {
      if (DD && p.LDD) {   // verticalmixing_mod.F90:103-108
        const double talpha = row(Q_DM)[k], sbeta = row(Q_S1)[k], T = row(Q_BET)[k], S = row(Q_S2)[k];
        row(Q_X1)[k] = 0.5 * (talpha + row(Q_DM)[k + 1]) * (T - row(Q_BET)[k + 1]);
        row(Q_X2)[k] = 0.5 * (sbeta + row(Q_S1)[k + 1]) * (S - row(Q_S2)[k + 1]);
      }
    }
    const double bk1 = aB[k + 1], uk1 = aU[k + 1], vk1 = aV[k + 1];
    const double dbloc = buoy - bk1;
    const double shsq = (U - uk1) * (U - uk1) + (V - vk1) * (V - vk1);
    const double zdiff = zmk - c_zm[k + 1];
    const double shs = shsq + 1.e-16;
    const double Rig = div_fast(dbloc * zdiff, shs, rcp_refine(shs));
    if (actz) { row(Q_GM)[k] = Rig; row(Q_YT)[k] = dbloc; }
    if (is1) row(Q_GM)[0] = 0.0;
    if (isnzp1) row(Q_GM)[k] = 0.0;
    if (p.diag && si[I_MAYBE]) {
      const size_t od = ro + k;
      if (actz) { if (p.LRI) p.Rig[od] = Rig; p.dbloc[od] = dbloc; p.Shsq[od] = shsq; }   // (Rig is rimix's, rimix_mod.F90:47-56)
    }
  };

  // =========================== persistent pass loop ===========================
#ifdef MCKPP_PS_STAMPS   // profiling build: per-segment cycle sums kept in registers by the manager wave
  unsigned long long tacc[31];
#pragma unroll
  for (int i = 0; i < 31; ++i) tacc[i] = 0;
  unsigned long long tlast = __builtin_amdgcn_s_memtime();
#define STAMP(i)                                              \
  do {                                                        \
    unsigned long long t_ = __builtin_amdgcn_s_memtime();     \
    tacc[i] += t_ - tlast;                                    \
    tlast = t_;                                               \
  } while (0)
#else
#define STAMP(i)
#endif
  // One iteration of the loop: a pass of the active slots (if any), the finish round (if some slot finishes), M0 (if
  // a slot finished, or one holds a ticket it is waiting to start: several steps in one launch).  M0 has this one
  // call site; the first iteration comes in as "no slot active, one waiting" and only fills the slots.
  bool first_iteration = true;
  for (;;) {
    const int any_active = first_iteration ? 0 : s_flags[0], any_waiting = first_iteration ? 1 : s_flags[6];
    if (!any_active && !any_waiting) break;
    int finishing = 0;
    // the view of this iteration (M0 and G_late change it between iterations only), and whether this pass ends in another
    const int goview = solo_perm ? 0 : __builtin_amdgcn_readfirstlane(s_flags[S_GOVIEW]);
    if (!solo_perm && sparse && __builtin_amdgcn_readfirstlane(s_flags[S_KVIEW]) == 0) set_view(0);   // (M0: its columns are done)
    if (any_active) {
    STAMP(22);
#ifdef MCKPP_PS_STAMPS
    tacc[23] += 1;
#endif

    // ---- L1: (new column / retry: ocnstep_mod.F90:91-112 extrapolation) under-relaxation, equation of state.
    // Items that had everything but V done while the V sweep ran (L1_ALL_BUT_V below) only relax V now.
    FOR_ITEMS
      const bool ahead = l1_ahead && si[I_L1A] && wv != mgr;
      L1_item(k, kr, si, my, sc, ro, act, virt1, virt2, is1, xs_, first_, ahead ? L1_V_ONLY : L1_FULL);
    END_ITEMS
    STAMP(0);
    __syncthreads();
    STAMP(1);
    if (goview > 0) {
      // From here on the workgroup works in a view of the `goview` slots of s_flags[S_MAP ..] (G_early of the pass
      // before).  Their rows and records stay where they are; what changes hands is the iterate of the under-relaxation,
      // which lives in the registers of the threads that own the slots' items: through the scratch rows of the iterate in
      // global memory (a block per (workgroup, slot): where the items of a third trip keep theirs).
      FOR_ITEMS
        if (!act) continue;
        if (first_ == 0) { xs_[0] = rU; xs_[LS] = rV; xs_[2 * LS] = rT; xs_[3 * LS] = rS; }
        else if (first_ == 1) { xs_[0] = r2U; xs_[LS] = r2V; xs_[2 * LS] = r2T; xs_[3 * LS] = r2S; }
      END_ITEMS
      __syncthreads();
      if (tid == 0) { s_flags[S_KVIEW] = goview; s_flags[S_GOVIEW] = 0; }
      set_view(goview);
      FOR_ITEMS
        if (!act) continue;
        rU = xs_[0]; rV = xs_[LS]; rT = xs_[2 * LS]; rS = xs_[3 * LS];
      END_ITEMS
    }

    // ---- L2a: the whole-layer terms of the reference-level sums (verticalmixing_mod.F90:122-127 with wz = dz,
    // del = 0.5), once per layer instead of once per (level, layer)
    if (p.l2pre) {   // (chosen by the host: deep reference-level sums, and no double diffusion - it has the two rows then)
      FOR_ITEMS
        if (!actz) continue;
        const double dzk = c_zm[k] - c_zm[k + 1];
        const double Uk = row(Q_YU)[k], Uk1 = row(Q_YU)[k + 1], Vk = row(Q_YV)[k], Vk1 = row(Q_YV)[k + 1];
        const double nu = dzk * (Uk + 0.5 * (Uk1 - Uk)), nv = dzk * (Vk + 0.5 * (Vk1 - Vk));
        row(Q_DM)[k] = nu; row(Q_BET)[k] = nv;
        if (__builtin_expect(tiny_nonzero(nu) || tiny_nonzero(nv), 0)) si[I_TINY] = 1;
      END_ITEMS
      __syncthreads();
    }
    // ---- M1 | L2: surface fluxes (wave 0) | reference-level loop, Ri pieces (verticalmixing_mod.F90:111-137)
    // The bulk Richardson numbers of bldepth, and with them the reference-level averages here, are formed down to
    // the level the scan of the pass before ended at plus eight (s_flags[3]; every level for a new column, with
    // double diffusion - whose rows the second round below would need - and in the modes that return uref/vref of
    // the deepest level).  MCKPP_L3_CAP caps the guess (tests).
    const int kguess = (DD || p.mode != MCKPP_MODE_STEP) ? nz : (p.l3cap > 0 && p.l3cap < s_flags[3] ? p.l3cap : s_flags[3]);
    if (wv == mgr) M1();
    if (kguess < nz && sparse) {   // a view of a few slots: one level-major item per thread, none on the manager wave
      FOR_ITEMS_RISING
        L2_item(k, si, my, sc, (size_t)si[I_COL] * p.ld, actz, is1, isnz, k == nzp1, k <= kguess);
      END_ITEMS
    } else if (kguess < nz) {
      // level-major order, rising: the waves that hold levels below the guess have the cheap part only
      // (all items to the waves other than the manager's, which has M1 to do - but for a last trip of a few items:
      // 915 items on 448 threads are two trips and nineteen items, which one wave would go a third time for while
      // the manager, done with M1, waits; it takes them)
      const int nfull = nthreads > 64 ? (nitems_lm / nhelp) * nhelp : nitems_lm, nrem = nitems_lm - nfull;
      const bool mgr_rest = nfull > 0 && nrem > 0 && nrem <= 64;
      const int nhelpers = mgr_rest ? nfull : nitems_lm;
      auto l2_one = [&](const int it_) {
        const int k = (W == 1 ? it_ : (int)__umulhi((unsigned)it_, Wmagic)) + 1;
        const int slot = it_ - (k - 1) * W;
        int *const si = sirec + slot * I_COUNT;
        if (!si[I_ACT]) return;
        L2_item(k, si, slots + slot * SS, screc + slot * C_COUNT, (size_t)si[I_COL] * p.ld, k <= nz, k == 1, k == nz, k == nzp1, k <= kguess);
      };
      for (int it_ = tid2 >= 0 ? tid2 : nhelpers; it_ < nhelpers; it_ += nhelp) l2_one(it_);
      if (wv == mgr && mgr_rest && lane < nrem) l2_one(nfull + lane);
    } else if (nzp1 >= 50) {
      // measured: the level-major order (a deep and a shallow item per thread) pays from ~50 levels on (+2 % at 60,
      // +13 % on the stretched 69-level grid, -2 % at 40)
      FOR_ITEMS_BY_LEVEL
        L2_item(k, si, my, sc, ro, actz, is1, isnz, isnzp1, true);
      END_ITEMS
    } else {
      FOR_ITEMS
        if (!act) continue;   // the two equation-of-state items exist for L1 only
        L2_item(k, si, my, sc, ro, actz, is1, isnz, isnzp1, true);
      END_ITEMS
    }
    STAMP(2);
    __syncthreads();
    STAMP(3);

    // ---- L3: rimix + z121 (rimix_mod.F90:13-106, z121_mod.F90:7-45), ddmix, interior diffusivity rows;
    //          bldepth, level-parallel part (bldepth_mod.F90:105-147) - the bulk Richardson numbers only down to
    //          kguess (they cost two thirds of the phase, and the scan will not look further unless the boundary
    //          layer has deepened: then the rest is formed after it, below).  Level-major order, rising: the waves
    //          that hold the deeper levels do the rimix part only.
    auto bulk_ri = [&](const int k, int *const si, double *const my, double *const sc, const double Ritop, const double dVsq,
                       const bool actz, const bool is1) {
      auto row = [&](int a) -> strided<ROWS> { return strided<ROWS>{my + a}; };
      const strided<ROWS> aDb = row(Q_YT);
      const double zmk = c_zm[k], zdiff = zmk - c_zm[k + 1], dbloc = aDb[k];
      const double B0 = sc[C_B0], B0sol = sc[C_B0SOL], ustar = sc[C_USTAR];
      wscale_u wu;
      wu.ju = si[I_JU]; wu.ufrac = sc[C_UFRAC]; wu.ustar = ustar; wu.ucube = sc[C_UCUBE];
      const double zm_kmp1 = c_zm[nzp1];
      double swf = p.swfrac_tab[si[I_JER] * p.ldc + k];
      double bf = B0 + B0sol * (1. - swf);
      double st = 0.5 + dsign(0.5, bf + epsln16);
      double sg = st * 1. + (1. - st) * eps01;
      double wm, ws;
      wscale_dev(p, wu, sg, -zmk, bf, wm, ws);
      double dbm1 = aDb[k - 1];
      double bvsq = 0.5 * (div_fast(dbm1, c_zm[k - 1] - zmk, c_rdz[k - 1]) + div_fast(dbloc, zdiff, c_rdz[k]));
      double Vtsq = -zmk * ws * __builtin_sqrt(__builtin_fabs(bvsq)) * p.Vtc;
      const double rawden = dVsq + Vtsq + epsln16, bfa = __builtin_fabs(bf) + epsln16;
      double raw = div_fast(Ritop, rawden, rcp_refine(rawden));
      double dmo = div_fast(div_fast(cmonob * ustar * ustar * ustar, p.vonk, c_misc[1]), bfa, rcp_refine(bfa));
      dmo = st * dmo - (1. - st) * zm_kmp1;
      if (k >= 2 && actz) { row(Q_YV)[k] = raw; row(Q_YU)[k] = dmo; }
      if (is1) { row(Q_YV)[1] = 0.0; row(Q_YU)[1] = -zm_kmp1; }
    };
    FOR_ITEMS_RISING
      const strided<ROWS> aR = row(Q_GM);
      const double Rig = aR[k];
      const double Riinfty = 0.8;
      double vm1 = aR[k - 1], vp1 = aR[k + 1];
      double wm1 = (k - 1 >= 1 && !((vm1 < 0.0) || (vm1 > Riinfty))) ? 1.0 : 0.0;
      double wp1 = (k + 1 <= nz && !((vp1 < 0.0) || (vp1 > Riinfty))) ? 1.0 : 0.0;
      double sm = wm1 * vm1 + 2. * Rig + wp1 * vp1;
      double wait = wm1 + 2.0 + wp1;
      sm = div_fast(sm, wait, wait == 3.0 ? 1. / 3. : (wait == 2.0 ? 0.5 : 0.25));
      double Rigg = dmax2(sm, 0.0);
      double ratio = dmin2(div_fast(Rigg, Riinfty, 1. / Riinfty), 1.0);
      double fri = (1.0 - ratio * ratio);
      fri = fri * fri * fri;
      double dm_i = (0.0001 + fri * 0.005);
      double ds_i = (0.00001 + fri * 0.005);
      double dt_i = ds_i;   // dift = difs, rimix_mod.F90:95-97
      if (!p.LRI) { dm_i = 0.0; ds_i = 0.0; dt_i = 0.0; }   // kppmix_mod.F90:65-74: zeroed, rimix not called
      if constexpr (EXT) {
        if (DD && p.LDD) {   // ddmix_mod.F90:12-52
          const double Rrho0 = 1.9, dsfmax = 1.0e-4;
          const double aDT = row(Q_X1)[k], bDS = row(Q_X2)[k];
          if ((aDT > bDS) && (bDS > 0.)) {
            double Rrho = dmin2(aDT / bDS, Rrho0);
            double rr = ((Rrho - 1) / (Rrho0 - 1));
            double diffdd = 1.0 - rr * rr;
            diffdd = dsfmax * diffdd * diffdd * diffdd;
            dt_i = dt_i + diffdd * 0.8 / Rrho;
            ds_i = ds_i + diffdd;
          } else if ((aDT < 0.0) && (bDS < 0.0) && (aDT < bDS)) {
            double Rrho = aDT / bDS;
            double diffdd = 1.5e-6 * 9.0 * 0.101 * mckpp_exp(4.6 * mckpp_exp(-0.54 * (1 / Rrho - 1)));
            double prandtl = 0.15 * Rrho;
            if (Rrho > 0.5) prandtl = (1.85 - 0.85 / Rrho) * Rrho;
            dt_i = dt_i + diffdd;
            ds_i = ds_i + prandtl * diffdd;
          }
        }
      }
      if (k <= kguess) bulk_ri(k, si, my, sc, row(Q_DT)[k], row(Q_DS)[k], actz, is1);   // (Ritop, dVsq: before dift takes the row)
      // interior diffusivities (after the reads of Ritop / dVsq, which share their rows)
      // (without double diffusion difs = dift bit for bit: one row, Q_DT, serves both from here on)
      if (actz) { row(Q_DM)[k] = dm_i; if (DD) row(Q_DS)[k] = ds_i; row(Q_DT)[k] = dt_i; }
      if (isnz) { row(Q_DM)[k + 1] = dm_i; if (DD) row(Q_DS)[k + 1] = ds_i; row(Q_DT)[k + 1] = dt_i; }   // kppmix_mod.F90:82-84
      if (is1) { row(Q_DM)[0] = 0.0; if (DD) row(Q_DS)[0] = 0.0; row(Q_DT)[0] = 0.0; }
    END_ITEMS
    STAMP(4);
    __syncthreads();
    STAMP(5);

    // ---- M2: Rib(ku) = MAX(Rib(ku), Rib(ka)+epsln), bldepth_mod.F90:137
    // (the scan runs down to the level L3 went to; if it gets there without every column having crossed Ricr -
    // the boundary layer has deepened since the pass the guess comes from - the other levels follow)
    auto scan_result = [&](int kmax) {   // what the scan's outcome means for L4, for a second round and for the next pass
      if (lane < W && sirec[lane * I_COUNT + I_ACT]) {
        const int kdone = s_flags[2], stopped = s_flags[5];
        if (stopped) { s_flags[4] = 0; s_flags[3] = kdone + 8 < nz ? kdone + 8 : nz; }
        else if (kmax >= nz) { s_flags[4] = 0; s_flags[3] = nz; }
        else s_flags[4] = 1;
      }
    };
    if (wv == mgr) {
      if (lane == 0) { s_flags[2] = nz; s_flags[4] = 0; s_flags[5] = 0; }
      // levels above the scan's start are taken as they are; the scan's fifth slot of s_flags: its `stopped`
      {
        int out[2] = {nz, 0};
        ps_scan_rib(W, Q_YV, slots, SS, ROWS, 2, kguess, sirec + I_ACT, I_COUNT, lane, Ricr, out);
        if (lane < W && sirec[lane * I_COUNT + I_ACT]) { s_flags[2] = out[0]; s_flags[5] = out[1]; }
      }
      scan_result(kguess);
    }
    STAMP(6);
    __syncthreads();
    if (s_flags[4]) {   // rare: what the levels below the guess were spared - reference averages, Ritop, dVsq, bulk
      // Richardson number - then the scan goes on.  The rows of the iterate's U and V hold the Monin-Obukhov depths
      // and Richardson numbers of the upper levels by now: copies from the registers / the scratch into two rows
      // that are free (Rig is used up, the sweeps' q not yet formed).
      FOR_ITEMS
        if (!act) continue;
        row(Q_GM)[k] = first_ == 0 ? rU : first_ == 1 ? r2U : xs_[0];
        row(Q_BET)[k] = first_ == 0 ? rV : first_ == 1 ? r2V : xs_[LS];
      END_ITEMS
      __syncthreads();
      FOR_ITEMS_RISING
        if (k > kguess && actz) {
          double ur, vr, br;
          ref_sums(k, si, my, Q_GM, Q_BET, false, actz, ur, vr, br);
          const double zmk = c_zm[k], zref = eps01 * zmk;
          const double U = row(Q_GM)[k], V = row(Q_BET)[k], buoy = row(Q_YS)[k];
          bulk_ri(k, si, my, sc, (zref - zmk) * (br - buoy), (ur - U) * (ur - U) + (vr - V) * (vr - V), actz, is1);
        }
      END_ITEMS
      __syncthreads();
      if (wv == mgr) {
        int out[2] = {nz, 0};
        ps_scan_rib(W, Q_YV, slots, SS, ROWS, kguess + 1, nz, sirec + I_ACT, I_COUNT, lane, Ricr, out);
        if (lane < W && sirec[lane * I_COUNT + I_ACT]) { s_flags[2] = out[0]; s_flags[5] = out[1]; }
        scan_result(nz);
      }
      __syncthreads();
    }
    STAMP(7);

    // ---- L4: first level with hmin < -zm(k) (bldepth_mod.F90:139-180): every hit level posts its hmin,
    //          the shallowest one wins through an LDS minimum.  Level-major order: the levels below the one the
    //          scan stopped at cannot be the first (ps_scan_rib), and the waves that hold only such levels have
    //          nothing to do - with the boundary layer in the upper third of the column the phase is one trip
    //          of the item loop on a few waves.
    FOR_ITEMS_RISING
      if (k > s_flags[2]) break;   // (a thread's later items are deeper still)
      const strided<ROWS> aRaw = row(Q_YV), aDmo = row(Q_YU);
      const double zmk = c_zm[k];
      const double ocdepth = sc[C_OCDEPTH];
      const double zm_kmp1 = c_zm[nzp1];
      const double hek = sc[C_HEK];
      const double B0 = sc[C_B0], B0sol = sc[C_B0SOL];
      double swf = p.swfrac_tab[si[I_JER] * p.ldc + k];
      double bf = B0 + B0sol * (1. - swf);
      double stab = 0.5 + dsign(0.5, bf + epsln16);
      double Rka = aRaw[k - 1], Rku = aRaw[k], dmoa = aDmo[k - 1], dmou = aDmo[k];
      double zkm1 = c_zm[k - 1];
      double hri = -zkm1 + (zkm1 - zmk) * (Ricr - Rka) / (Rku - Rka);
      double hmonob;
      if (dmou <= (-zmk)) {
        hmonob = (dmou - dmoa) / (zkm1 - zmk);
        hmonob = (dmou + hmonob * zmk) / (1. - hmonob);
      } else {
        hmonob = -zm_kmp1;
      }
      double hekman = stab * hek - (1. - stab) * zm_kmp1;
      double hmin = dmin2(dmin2(dmin2(hri, hmonob), hekman), -ocdepth);
      bool hit = (k >= 2) && actz && (hmin < -zmk);
      if (hit && !si[I_INITFLAG] && (hmin < -zkm1)) {
        double hmin2 = dmin2(dmin2(hri, hmonob), -ocdepth);
        if (hmin2 < -zmk) hmin = hmin2;
      }
      if (hit) { row(Q_YS)[k] = hmin; atomicMin(&si[I_KBLC], k); }
    END_ITEMS
    STAMP(8);
    __syncthreads();
    STAMP(9);

    // ---- M3: hbl, kbl, slot-uniform part of blmix
    // The right-hand side of U (ocnint_mod.F90:51-54, tridrhs solvers.F90:53-107) needs nothing of this pass's mixing:
    // the old time level, the iterate's V, the surface stress.  The waves other than the manager's form it for their
    // items here, while M3 runs (the row is free: the Monin-Obukhov depths are used up); the manager's own items get
    // theirs in L6.
    auto rhs_u = [&](int k, bool actz, bool isnzp1, size_t ro, const double *sc, double *my, int first_, auto xs_) {
      const size_t o = ro + (k - 1);
      const double Uo = p.U[o];
      if (actz) {
        const double Vo = p.V[o];
        double Ubot = 0.0;
        if (k == nz) Ubot = p.U[ro + (nzp1 - 1)];
        const double V = first_ == 0 ? rV : first_ == 1 ? r2V : xs_[LS];   // of the iterate
        const double dto = p.dto, f = sc[C_F];
        double rhsU;
        if (k == 1) rhsU = Uo + dto * (f * .5 * (Vo + V) - div_fast(sc[C_WU01], c_hm[1], c_misc[0]));
        else rhsU = Uo + dto * f * .5 * (Vo + V);
        if (k == nz) rhsU = rhsU + c_t1[nz] * 0.0001 /* difm(nz): kppmix / L5 */ * Ubot;
        my[k * ROWS + Q_YU] = rhsU;
      }
      if (isnzp1) my[k * ROWS + Q_YU] = Uo;   // solvers.F90:159
    };
    if (wv == mgr) M3();   // (the ocnstep control, G_early, follows behind the manager's own L6 items, where it would wait)
    else if (do_ocnint) {
      FOR_ITEMS
        if (!act) continue;
        rhs_u(k, actz, isnzp1, ro, sc, my, first_, xs_);
      END_ITEMS
    }
    STAMP(10);
    __syncthreads();
    STAMP(11);

    // ---- L5: blmix shape functions, enhance, combine (blmix_mod.F90:110-133, enhance_mod.F90:10-51,
    //          kppmix_mod.F90:103-111, verticalmixing_mod.F90:151-159) -> final diffusivity rows
    //          (level-major order, rising: the shape functions and their table look-up are for the levels above
    //          kbl only - the waves that hold deeper levels just pass the interior values on)
    FOR_ITEMS_RISING
      const size_t ro = (size_t)si[I_COL] * p.ld;
      const int kbl = si[I_KBL];
      const double zmk = c_zm[k];
      const double dm_i = row(Q_DM)[k], dt_l = row(Q_DT)[k], ds_i = DD ? row(Q_DS)[k] : dt_l;   // interior values of L3
      double difm = dm_i, difs = ds_i, dift = dt_l, ghat = 0.;
      if (k < kbl) {
        const double hbl = sc[C_HBL], r_hbl = sc[C_RHBL], stable = sc[C_STABLE], bfsfc = sc[C_BFSFC];
        wscale_u wu;
        wu.ju = si[I_JU]; wu.ufrac = sc[C_UFRAC]; wu.ustar = sc[C_USTAR]; wu.ucube = sc[C_UCUBE];
        const double hk = c_hm[k];
        double wm, ws;
        double sig = div_fast(-zmk + 0.5 * hk, hbl, r_hbl);
        double sigma = stable * sig + (1. - stable) * dmin2(sig, eps01);
        wscale_dev(p, wu, sigma, hbl, bfsfc, wm, ws);
        double a1 = sig - 2.;
        double a2 = 3. - 2. * sig;
        double a3 = sig - 1.;
        double Gm = a1 + a2 * sc[C_GAT1 + 0] + a3 * sc[C_DAT1 + 0];
        double Gt = a1 + a2 * sc[C_GAT1 + 2] + a3 * sc[C_DAT1 + 2];
        double b0 = hbl * wm * sig * (1. + sig * Gm);
        double b2 = hbl * ws * sig * (1. + sig * Gt);
        double b1 = b2;   // same operands, same operations unless double diffusion separates difs from dift
        if (DD) {
          double Gs = a1 + a2 * sc[C_GAT1 + 1] + a3 * sc[C_DAT1 + 1];
          b1 = hbl * ws * sig * (1. + sig * Gs);
        }
        const double ghd = ws * hbl + epsln20;
        double gh = div_fast((1. - stable) * p.cg, ghd, rcp_refine(ghd));
        if (k == kbl - 1 && k <= nz - 1) {
          const double caseA = sc[C_CASEA];
          double delta = div_fast(hbl + zmk, zmk - c_zm[k + 1], c_rdz[k]);
          double omd = 1. - delta;
          double dkmp5 = caseA * dm_i + (1. - caseA) * b0;
          double dstar = (omd * omd) * sc[C_DKM1 + 0] + (delta * delta) * dkmp5;
          b0 = omd * dm_i + delta * dstar;
          if (DD) {
            dkmp5 = caseA * ds_i + (1. - caseA) * b1;
            dstar = (omd * omd) * sc[C_DKM1 + 1] + (delta * delta) * dkmp5;
            b1 = omd * ds_i + delta * dstar;
          }
          dkmp5 = caseA * dt_l + (1. - caseA) * b2;
          dstar = (omd * omd) * sc[C_DKM1 + 2] + (delta * delta) * dkmp5;
          b2 = omd * dt_l + delta * dstar;
          if (!DD) b1 = b2;
          gh = (1. - caseA) * gh;
        }
        difm = b0; difs = b1; dift = b2; ghat = gh;
      }
      if (k >= nz) { difm = 0.0001; difs = 0.00001; dift = 0.00001; ghat = 0.0; }
      // what the sweeps take: p(k) = tri(k,1) diff(k), q(k+1) = tri(k+1,0) diff(k) (ps_sysrows); L6: dift, difs, ghat
      if (k <= nz) {
        const double t1k = c_t1[k];
        row(Q_DM)[k] = t1k * difm; row(Q_DT)[k] = t1k * dift; if (DD) row(Q_DS)[k] = t1k * difs;
        if (k < nz) {
          const double t0n = c_t0[k + 1];
          row(Q_GM)[k + 1] = t0n * difm; row(Q_BET)[k + 1] = t0n * dift; if (DD) row(Q_S1)[k + 1] = t0n * difs;
        }
      }
      row(ps_sysrows<XV>::dl6_t)[k] = dift; if (DD) row(ps_sysrows<XV>::dl6_s)[k] = difs; row(Q_YV)[k] = ghat;
      if (p.diag && si[I_MAYBE]) {   // the sweeps reuse these rows: what the last vmix leaves behind goes out now
        const size_t od = ro + k;
        p.difm[od] = difm; p.difs[od] = difs; p.dift[od] = dift;
        if (actz) p.ghat[od] = ghat;
      }
    END_ITEMS
    STAMP(12);
    __syncthreads();
    STAMP(13);

    // ---- L6: right-hand sides of U, T, S (ocnint_mod.F90:51-58, tridrhs solvers.F90:53-107)
    if (do_ocnint) {
      FOR_ITEMS
        if (!act) continue;
        const strided<ROWS> aDt = row(ps_sysrows<XV>::dl6_t), aDs = row(ps_sysrows<XV>::dl6_s), aGh = row(Q_YV);
        const size_t o = ro + (k - 1);
        const double To = p.T[o], So = p.S[o];
        if (wv == mgr) rhs_u(k, actz, isnzp1, ro, sc, my, first_, xs_);   // (the others': under M3)
        const double tri1_nz = c_t1[nz];
        const double wX0_1 = sc[C_WX01], wX0_2 = sc[C_WX02];
        const strided<ROWS> yT = row(Q_YT), yS = row(Q_YS);
        if (actz) {
          const double difs = aDs[k], dift = aDt[k], ghat = aGh[k];
          const int jer = si[I_JER];
          const double rho0cp0 = sc[C_RHO0CP0], r_rc = sc[C_RRC], sflux3 = sc[C_SFLUX3];
          const double dt_m1 = aDt[k - 1], ds_m1 = aDs[k - 1];
          const double gh_m1 = (k >= 2) ? aGh[k - 1] : 0.0;
          double wxnt = 0.0, wxnt_m1 = 0.0;   // wXNT(k,1), wXNT(k-1,1), fluxes_mod.F90:110-116
          if (ntime >= 1) {
            wxnt = div_fast(-sflux3 * p.swdk_tab[jer * p.ldc + k], rho0cp0, r_rc);
            wxnt_m1 = div_fast(-sflux3 * p.swdk_tab[jer * p.ldc + k - 1], rho0cp0, r_rc);
          }
          double rhsT;
          const double dtohk = c_dtohk[k];
          if (k == 1) rhsT = To + dtohk * (wX0_1 * dift * ghat - wX0_1 * 1.0 + wxnt - sc[C_WXNT0]);
          else rhsT = To + dtohk * (wX0_1 * (dift * ghat - dt_m1 * gh_m1) + wxnt - wxnt_m1);
          if (k == nz && nz > 1) rhsT = rhsT + p.T[ro + (nzp1 - 1)] * tri1_nz * dift;
          double rhsS;
          if (k == 1) rhsS = So + dtohk * (wX0_2 * difs * ghat - wX0_2 * 1.0 + 0.0 - 0.0);
          else rhsS = So + dtohk * (wX0_2 * (difs * ghat - ds_m1 * gh_m1) + 0.0 - 0.0);
          if (k == nz && nz > 1) rhsS = rhsS + p.S[ro + (nzp1 - 1)] * tri1_nz * difs;
          if constexpr (EXT) ext_rhs(my, si, col, k, si[I_KBL], To, So, rhsT, rhsS);
          yT[k] = rhsT; yS[k] = rhsS;
        }
        if (isnzp1) {
          yT[k] = To; yS[k] = So;   // solvers.F90:159
          if constexpr (EXT) { double t = 0.0, s2 = 0.0; ext_rhs(my, si, col, k, si[I_KBL], To, So, t, s2); }   // ocnint_mod.F90:153-160, 207-213
        }
      END_ITEMS
    }
    // The ocnstep control of this pass: nothing in L5 / L6 reads what it writes, and here the manager wave - one trip of
    // items where the others have two - would only wait at the barrier.
    if (wv == mgr) G_early();
    STAMP(14);
    __syncthreads();
    STAMP(15);

    // ---- M4: Thomas factorise + sweep for U, T, S (solvers.F90:14-44, 112-161)
    if constexpr (SM == 0) {
      if (wv == mgr && do_ocnint) {
        if constexpr (XV != 2 && PS_FWD4) ps_thomas_uts_fwd4<XV>(W, slots, SS, nz, sirec + I_ACT, I_COUNT, sirec + I_BAD, I_COUNT, lane);
        else ps_thomas_uts_fwd<XV>(W, slots, SS, nz, sirec + I_ACT, I_COUNT, sirec + I_BAD, I_COUNT, lane);
        STAMP(24);
        ps_thomas_uts_back<XV>(W, slots, SS, nz, sirec + I_ACT, I_COUNT, lane);
      }
    } else if (do_ocnint) {
      // solver mode 1: the upper half of every system on the manager wave, the lower half on another wave
      // (a workgroup of one wave: one after the other); the 2x2 system in the middle needs both
      // Which wave: the hardware deals the waves of the two 8-wave workgroups of a CU to SIMDs 2,1,3,0,2,1,3,0 and
      // 1,3,0,2,1,3,0,2 (tools/ubench/hwid.hip) - wave 1 of the first shares a SIMD with the manager wave of the second,
      // wave 2 does not (+0.5 ... 0.7 % at 60 and 69 levels; with one 16-wave workgroup per CU wave 1 is the better one)
      const int wv2 = nthreads == 512 ? 2 : nthreads > 64 ? 1 : 0;
      if constexpr (XV != 2 && PS_FWD4) {
        if (nz >= 16) {
          if (wv == mgr) ps_thomas2_uts_fwd4<XV, 1>(W, slots, SS, nz, sirec + I_ACT, I_COUNT, sirec + I_BAD, I_COUNT, lane);
          if (wv == wv2) ps_thomas2_uts_fwd4<XV, -1>(W, slots, SS, nz, sirec + I_ACT, I_COUNT, sirec + I_BAD, I_COUNT, lane);
        } else {
          if (wv == mgr) ps_thomas2_uts_fwd<XV, 1>(W, slots, SS, nz, sirec + I_ACT, I_COUNT, sirec + I_BAD, I_COUNT, lane);
          if (wv == wv2) ps_thomas2_uts_fwd<XV, -1>(W, slots, SS, nz, sirec + I_ACT, I_COUNT, sirec + I_BAD, I_COUNT, lane);
        }
      } else {
        if (wv == mgr) ps_thomas2_uts_fwd<XV, 1>(W, slots, SS, nz, sirec + I_ACT, I_COUNT, sirec + I_BAD, I_COUNT, lane);
        if (wv == wv2) ps_thomas2_uts_fwd<XV, -1>(W, slots, SS, nz, sirec + I_ACT, I_COUNT, sirec + I_BAD, I_COUNT, lane);
      }
      STAMP(24);
      __syncthreads();
      if (wv == mgr) ps_thomas2_uts_back<XV, 1>(W, slots, SS, nz, sirec + I_ACT, I_COUNT, sirec + I_BAD, I_COUNT, lane);
      if (wv == wv2) ps_thomas2_uts_back<XV, -1>(W, slots, SS, nz, sirec + I_ACT, I_COUNT, sirec + I_BAD, I_COUNT, lane);
    }
    STAMP(16);
    __syncthreads();
    STAMP(17);

    // ---- L7: V right-hand side with the new U (ocnint_mod.F90:62-69)
    if (do_ocnint) {
      FOR_ITEMS
        if (!act) continue;
        const size_t o = ro + (k - 1);
        const double Uo = p.U[o], Vo = p.V[o];
        const double dto = p.dto, f = sc[C_F];
        const strided<ROWS> yU = row(Q_YU), yV = row(Q_YV);
        if (actz) {
          // beside the right-hand side, what the V sweep needs of the momentum factorisation apart from the
          // recurrence itself: the pivot and its refined reciprocal, exactly the values the U sweep formed (same
          // operations on the same p, q, gam; solvers.F90:140-151's replacement of a zero pivot included), into
          // two rows the T and S systems are done with
          const double pk = row(Q_DM)[k], qk = row(Q_GM)[k], gk = row(ps_sysrows<XV>::gam_m)[k];
          double betk;
          if constexpr (SM == 0) {
            betk = (k == 1) ? 1. + pk : ((1. + pk) + qk) + qk * gk;
          } else {
            // solver mode 1: the pivots of either half as its sweep formed them (ps_thomas2_uts_fwd), and the V sweep's
            // other per-level operands where both of its directions look for them (ps_thomas2_v): the multiplier of
            // the neighbour's solution (q(k) stays | p(k)) and the multiplier of the substitution (gam(k+1) | g(k))
            const int m = nz >> 1;
            const double gk1 = row(ps_sysrows<XV>::gam_m)[k + 1];
            if (k <= m) {
              betk = (k == 1) ? 1. + pk : ((1. + pk) + qk) + qk * gk;
              row(Q_DM)[k] = gk1;
            } else {
              betk = (k == nz) ? (1. + pk) + qk : ((1. + pk) + qk) + pk * gk1;
              row(Q_GM)[k] = pk;
              row(Q_DM)[k] = gk;
            }
          }
          if (betk == 0.) betk = 1.E-12;
          row(Q_BET)[k] = betk;
          row(Q_DT)[k] = rcp_refine(betk);
          const double un = yU[k];
          double rhsV;
          if (k == 1) rhsV = Vo - dto * (f * .5 * (Uo + un) + div_fast(sc[C_WU02], c_hm[1], c_misc[0]));
          else rhsV = Vo - dto * f * .5 * (Uo + un);
          if (k == nz) rhsV = rhsV + c_t1[nz] * 0.0001 /* difm(nz): kppmix / L5 */ * p.V[ro + (nzp1 - 1)];
          yV[k] = rhsV;
        } else {
          yV[k] = Vo;
        }
      END_ITEMS
    }
    STAMP(18);
    __syncthreads();
    STAMP(19);

    // ---- M5: Thomas sweep for V on the stored momentum factorisation | the next pass's L1 but for V
    // (solver mode 1's V sweep can flag its middle pivot: there the bookkeeping stays behind it)
    const bool late_on_mgr = SM != 0 || nthreads <= 64;
    if (wv == mgr) {
      if (do_ocnint) {
        if constexpr (SM == 0) {
          ps_thomas_v_fwd(W, slots, SS, ROWS, nz, sirec + I_ACT, I_COUNT, lane);
          STAMP(25);
          ps_thomas_v_back(W, slots, SS, ROWS, ps_sysrows<XV>::gam_m, nz, sirec + I_ACT, I_COUNT, lane);
        } else {
          ps_thomas2_v(W, slots, SS, ROWS, ps_sysrows<XV>::gam_m, nz, sirec + I_ACT, I_COUNT, sirec + I_BAD, I_COUNT, lane);
        }
      }
      if (late_on_mgr) G_late();
    } else {
      // The end-of-pass bookkeeping (G_late) runs here, on the workgroup's last wave, under the V sweep - not behind it on
      // the manager wave, where its LDS round trips were 1.2 k cycles of every pass: nothing it touches (the status word,
      // the zero-pivot flag of M4, the tiny-term flag of L2a, this pass's `may be the last` flag) is read or written by
      // the V sweep in the reference-order mode or by the L1 that runs beside it.
      if (!late_on_mgr && wv == (nthreads >> 6) - 1) G_late();
      if (l1_ahead) {   // meanwhile: the next pass's L1, all but V, for the items of slots that go on iterating
      FOR_ITEMS
        if (!si[I_L1A]) continue;
        L1_item(k, kr, si, my, sc, ro, act, virt1, virt2, is1, xs_, first_, L1_ALL_BUT_V);
      END_ITEMS
      }
    }
    STAMP(20);
    __syncthreads();
    STAMP(21);
    finishing = s_flags[1];
    if (finishing) {

    // =========================== finish round ===========================
    // instability trap (ocnstep_mod.F90:200-236), then retry or outputs + check_profile.  The profiles of a
    // finishing slot are what the last ocnint returned: they stay in the slot's solution rows (Q_YU, Q_YV,
    // Q_YT, Q_YS) and every sub-phase works on them in place.  INIT / VMIX have no solution: U,V,T,S are the
    // column's own rows in HBM.
    // (one phase: the terms of the rms differences are formed whether or not some level violates the bounds; the
    // decision below looks at them only if none did)
    FOR_ITEMS
      if (!act) continue;   // the two equation-of-state items exist for L1 only
      if (si[I_FIN] != F_TRAP) continue;   // :200-219
      const size_t o = ro + (k - 1);
      const double U = row(Q_YU)[k], V = row(Q_YV)[k], T = row(Q_YT)[k], S = row(Q_YS)[k], tk1 = row(Q_YT)[k + 1];
      const double Uo = p.U[o], Vo = p.V[o], To = p.T[o], So = p.S[o];
      const bool v = actz && (__builtin_fabs(U) >= 10 || __builtin_fabs(V) >= 10 || __builtin_fabs(T - tk1) >= 10);
      if (v) atomicAdd(&si[I_NVIOL], 1);
      const double hk = c_hm[k];
      row(Q_DM)[k] = (U - Uo) * (U - Uo) * hk / p.dm_nz;
      row(Q_DT)[k] = (V - Vo) * (V - Vo) * hk / p.dm_nz;
      row(Q_DS)[k] = (T - To) * (T - To) * hk / p.dm_nz;
      row(Q_GM)[k] = (S - So) * (S - So) * hk / p.dm_nz;
    END_ITEMS
    __syncthreads();
    STAMP(26);
    if (wv == mgr) {   // trap decision, one lane per (slot, profile) for the rmsd sums, then one per slot
      if (lane < W) sirec[lane * I_COUNT + I_NOVER] = 0;
      for (int l = lane; l < 4 * W; l += 64) {
        const int ms = l >> 2, mq = l & 3;
        int *msi = sirec + ms * I_COUNT;
        if (msi[I_ACT] && msi[I_FIN] == F_TRAP && msi[I_NVIOL] == 0) {
          const strided<ROWS> t{slots + ms * SS + (mq == 3 ? (int)Q_GM : (int)Q_DM + mq)};
          double sum = 0.;   // in the reference's order; eight terms fetched at a time, then added one by one
          int q = 1;
          for (; q + 7 <= nzp1; q += 8) {
            const double a0 = t[q], a1 = t[q + 1], a2 = t[q + 2], a3 = t[q + 3], a4 = t[q + 4], a5 = t[q + 5], a6 = t[q + 6], a7 = t[q + 7];
            sum = sum + a0; sum = sum + a1; sum = sum + a2; sum = sum + a3; sum = sum + a4; sum = sum + a5; sum = sum + a6; sum = sum + a7;
          }
          for (; q <= nzp1; ++q) sum = sum + t[q];
          sum = __builtin_sqrt(sum);
          if (sum >= 1.0) atomicAdd(&msi[I_NOVER], 1);
        }
      }
      if (lane < W) {
        int *msi = sirec + lane * I_COUNT;
        double *msc = screc + lane * C_COUNT;
        if (msi[I_ACT] && msi[I_FIN] == F_TRAP) {
          int comp_flag = 0, status = msi[I_STATUS], nreset = msi[I_NRESET];
          double f = msc[C_F];
          const int nviol = msi[I_NVIOL];
          if (nviol > 0) {
            comp_flag = 1;
            for (int i = 0; i < nviol; ++i) f = f * 1.01;
          } else {
            const int nover = msi[I_NOVER];
            if (nover > 0) {
              comp_flag = 1;
              for (int i = 0; i < nover; ++i) f = f * 1.01;
            }
          }
          if (comp_flag) { status |= 4; msc[C_F] = f; }
          nreset = nreset + 1;
          if (nreset > 10) status |= 8;
          msi[I_COMP] = comp_flag; msi[I_STATUS] = status; msi[I_NRESET] = nreset;
          if (comp_flag && nreset <= 10) {   // retry, ocnstep_mod.F90:89
            msi[I_FIN] = F_NONE; msi[I_LOAD] = 2; msi[I_NPASS_TRY] = 0; msi[I_ICONV] = 0; msi[I_MAYBE] = 0;
          } else {
            msi[I_FIN] = F_FINAL;
            msi[I_NU] = 0; msi[I_NV] = 0; msi[I_NF] = 0;
          }
        }
      }
    }
    __syncthreads();
    STAMP(27);
    // ---- outputs.  Diagnostic fluxes (ocnstep_mod.F90:242-256 / initialize_ocean.F90:66-81) from the final
    // profiles and their k+1 neighbours; STEP: level-1 references, then (optional physics) current damping.
    FOR_ITEMS
      if (!act) continue;   // the two equation-of-state items exist for L1 only
      if (si[I_FIN] != F_FINAL) continue;
      const bool fstep = p.mode == MCKPP_MODE_STEP;
      const bool sol = do_ocnint;   // profiles in the solution rows
      const size_t o = ro + (k - 1);
      double U = 0, V = 0, T = 0, S = 0, uk1 = 0, vk1 = 0, tk1 = 0, sk1 = 0;
      if (act) {
        if (sol) { U = row(Q_YU)[k]; V = row(Q_YV)[k]; T = row(Q_YT)[k]; S = row(Q_YS)[k]; }
        else { U = p.U[o]; V = p.V[o]; T = p.T[o]; S = p.S[o]; }
      }
      if (actz && p.diag && flux_diag) {
        if (sol) { uk1 = row(Q_YU)[k + 1]; vk1 = row(Q_YV)[k + 1]; tk1 = row(Q_YT)[k + 1]; sk1 = row(Q_YS)[k + 1]; }
        else { uk1 = p.U[o + 1]; vk1 = p.V[o + 1]; tk1 = p.T[o + 1]; sk1 = p.S[o + 1]; }
      }
      if (p.diag) {
        const double wX0_1 = sc[C_WX01], wX0_2 = sc[C_WX02];
        const double rho0cp0 = sc[C_RHO0CP0], sflux3 = sc[C_SFLUX3];
        const size_t od = ro + k;
        {
          if (actz) {
            const double dfm = p.difm[od], dfs = p.difs[od], dft = p.dift[od], gh = p.ghat[od];   // stored by L5 of this pass
            // (every load of the item before its first store: one memory round trip, not two)
            const double talpha = flux_diag ? p.talpha[od] : 0.0, sbeta = flux_diag ? p.sbeta[od] : 0.0;   // of the last vmix (L1 of this pass)
            const double swdk = (ntime >= 1) ? p.swdk_tab[si[I_JER] * p.ldc + k] : 0.0;
            p.wXNT1[od] = (ntime >= 1) ? -sflux3 * swdk / rho0cp0 : 0.0;
            if (flux_diag) {
              double deltaz = 0.5 * (c_hm[k] + c_hm[k + 1]);
              double wX1 = -dfs * ((T - tk1) / deltaz - gh * wX0_1);
              double wX2 = -dfs * ((S - sk1) / deltaz - gh * wX0_2);
              if (p.LDD) wX1 = -dft * ((T - tk1) / deltaz - gh * wX0_1);
              p.wX1[od] = wX1; p.wX2[od] = wX2;
              p.wX3[od] = p.grav * (talpha * wX1 - sbeta * wX2);
              p.wU1[od] = -dfm * (U - uk1) / deltaz;
              p.wU2[od] = -dfm * (V - vk1) / deltaz;
            }
          }
        }
        if (is1) {   // index-0 entries
          p.difm[ro] = 0.0; p.difs[ro] = 0.0; p.dift[ro] = 0.0;
          p.wU1[ro] = sc[C_WU01]; p.wU2[ro] = sc[C_WU02];
          p.wX1[ro] = wX0_1; p.wX2[ro] = wX0_2; p.wX3[ro] = -sc[C_B0];
          p.wXNT1[ro] = sc[C_WXNT0];
        }
      }
      if (fstep && is1) {   // level-1 references, before any override touches the profiles
        const auto cs = p.cs + (size_t)col * MCKPP_CS;
        cs[CS_UREF] = U; cs[CS_VREF] = V; cs[CS_TREF] = T;
        cs[CS_SSURF] = p.L_SSref ? cs[CS_SSREF] : S + sc[C_SREF];
      }
      if constexpr (EXT) {
        if (fstep && p.L_DAMP_CURR && act) {   // ocnstep_mod.F90:317-340
          const double rr = (double)p.dt_uvdamp * (86400. / p.dto);
          double a = 0.99 * __builtin_fabs(U), b = (U * U) / rr;
          if (b < a) atomicAdd(&si[I_NU], 1);
          row(Q_YU)[k] = U - dsign(dmin2(a, b), U);
          a = 0.99 * __builtin_fabs(V); b = (V * V) / rr;
          if (b < a) atomicAdd(&si[I_NV], 1);
          row(Q_YV)[k] = V - dsign(dmin2(a, b), V);
        }
      }
    END_ITEMS
    __syncthreads();   // the k+1 neighbours above are read from the rows that check_profile rewrites below
    STAMP(28);
    // ---- STEP: new time level, check_profile (overrides.F90:42-125); the optional parts count over all
    // levels of the column: LDS counters between workgroup barriers (EXT build).
    FOR_ITEMS
      if (!act) continue;   // the two equation-of-state items exist for L1 only
      if (!(si[I_FIN] == F_FINAL && p.mode == MCKPP_MODE_STEP)) continue;
      const size_t o = ro + (k - 1);
      const int newi = 1 - si[I_NEW];   // old = new; new = 1 - old
      double U = 0, V = 0, T = 0, S = 0;
      if (act) { U = row(Q_YU)[k]; V = row(Q_YV)[k]; T = row(Q_YT)[k]; S = row(Q_YS)[k]; }
      if (act) { p.Us[newi][o] = U; p.Vs[newi][o] = V; p.Ts[newi][o] = T; p.Ss[newi][o] = S; }
      if (si[I_COMP] && act) {   // overrides.F90:57-78
        if (p.clim_present) { T = p.ocnT_clim[o]; S = p.sal_clim[o]; row(Q_YT)[k] = T; row(Q_YS)[k] = S; }
        U = p.U_init[o]; V = p.V_init[o];
        row(Q_YU)[k] = U; row(Q_YV)[k] = V;
      }
      if constexpr (EXT) {
        if (si[I_LOCEAN] && p.L_NO_FREEZE && act) {   // :85-94
          double xt = p.tinc_fcorr[ro + k];
          if (T < -1.8) { xt = xt + (-1.8 - T); T = -1.8; row(Q_YT)[k] = T; atomicAdd(&si[I_NF], 1); }
          p.tinc_fcorr[ro + k] = xt;
        }
      }
    END_ITEMS
    if constexpr (EXT) {
      __syncthreads();
      // isotherm check (:102-120): |T(k) - T(k-1)| dz and dz into the gam rows, summed by every item of the slot
      FOR_ITEMS
        if (!act) continue;   // the two equation-of-state items exist for L1 only
        if (!(si[I_FIN] == F_FINAL && p.mode == MCKPP_MODE_STEP && si[I_LOCEAN] && p.L_NO_ISOTHERM)) continue;
        if (k >= 2 && act) {
          const double dz = c_zm[k] - c_zm[k - 1];
          row(Q_DM)[k] = __builtin_fabs((row(Q_YT)[k] - row(Q_YT)[k - 1])) * dz;
          row(Q_DT)[k] = dz;
        }
      END_ITEMS
      __syncthreads();
    }
    FOR_ITEMS
      if (!act) continue;   // the two equation-of-state items exist for L1 only
      if (si[I_FIN] != F_FINAL) continue;
      const auto cs = p.cs + (size_t)col * MCKPP_CS;
      const auto ci = p.ci + (size_t)col * MCKPP_CI;
      const size_t o = ro + (k - 1);
      if (p.mode == MCKPP_MODE_STEP) {
        double reset_out = (double)si[I_NRESET];
        if (si[I_COMP]) reset_out = 999.;
        bool iso_reset = false;
        if constexpr (EXT) {
          if (si[I_LOCEAN] && p.L_NO_ISOTHERM) {
            const strided<ROWS> tD = row(Q_DM), tZ = row(Q_DT);
            double dtdz_total = 0., dz_total = 0.;
            for (int q = 2; q <= p.iso_bot; ++q) {
              dtdz_total = dtdz_total + tD[q];
              dz_total = dz_total + tZ[q];
            }
            dtdz_total = dtdz_total / dz_total;
            if (__builtin_fabs(dtdz_total) < p.iso_thresh) {
              iso_reset = true;
              reset_out = (-1.) * reset_out;
            }
          } else {
            reset_out = 0.0;   // :121-123
          }
        } else {
          reset_out = 0.0;     // :121-123 (no isotherm check in the default physics)
        }
        if (act) {
          double U = row(Q_YU)[k], V = row(Q_YV)[k], T = row(Q_YT)[k], S = row(Q_YS)[k];
          if (EXT && iso_reset) { T = p.ocnT_clim[o]; S = p.sal_clim[o]; }
          p.U[o] = U; p.V[o] = V; p.T[o] = T; p.S[o] = S;
        }
        if (is1) {
          const int old = si[I_NEW], newi = 1 - old;
          double dampu = 0.0, dampv = 0.0;
          if constexpr (EXT) {
            if (p.L_DAMP_CURR) {
              const double inc = 1.0 / (double)nzp1;
              const int nu = si[I_NU], nv = si[I_NV];
              for (int i = 0; i < nu; ++i) dampu = dampu + inc;
              for (int i = 0; i < nv; ++i) dampv = dampv + inc;
            }
            double freeze = cs[CS_FREEZE];
            if (si[I_LOCEAN] && p.L_NO_FREEZE) {
              const double inc = 1.0 / (double)nzp1;
              const int nf = si[I_NF];
              for (int i = 0; i < nf; ++i) freeze = freeze + inc;
            }
            cs[CS_FREEZE] = freeze;
          }
          const double hmixn = sc[C_HMIXN];
          cs[CS_HMIX] = hmixn;
          cs[CS_KMIX] = (double)si[I_KMIXN];
          cs[newi ? CS_HMIXD1 : CS_HMIXD0] = hmixn;
          cs[CS_RESET] = reset_out;
          cs[CS_DAMPU] = dampu; cs[CS_DAMPV] = dampv;
          ci[CI_OLD] = old; ci[CI_NEW] = newi;
          ci[CI_STATUS] = si[I_STATUS]; ci[CI_NPASS] = si[I_NPASS];
        }
      } else if (p.mode == MCKPP_MODE_INIT) {
        if (act) {
          const double U = p.U[o], V = p.V[o], T = p.T[o], S = p.S[o];
          p.Us[0][o] = U; p.Us[1][o] = U; p.Vs[0][o] = V; p.Vs[1][o] = V;
          p.Ts[0][o] = T; p.Ts[1][o] = T; p.Ss[0][o] = S; p.Ss[1][o] = S;
        }
        if (is1) {
          const double hbl = sc[C_HBL];
          cs[CS_HMIX] = hbl;
          cs[CS_KMIX] = (double)si[I_KBL];
          cs[CS_TREF] = p.T[o];
          cs[CS_UREF] = sc[C_UREFNZ]; cs[CS_VREF] = sc[C_VREFNZ];
          cs[CS_HMIXD0] = hbl; cs[CS_HMIXD1] = hbl;
          ci[CI_OLD] = 0; ci[CI_NEW] = 1; ci[CI_INITFLAG] = 0;
          ci[CI_STATUS] = si[I_STATUS]; ci[CI_NPASS] = si[I_NPASS];
        }
      } else {
        if (p.mode == MCKPP_MODE_PASS && act) {
          p.U[o] = row(Q_YU)[k]; p.V[o] = row(Q_YV)[k]; p.T[o] = row(Q_YT)[k]; p.S[o] = row(Q_YS)[k];
        }
        if (is1) {
          cs[CS_HMIX] = sc[C_HBL];
          cs[CS_KMIX] = (double)si[I_KBL];
          cs[CS_UREF] = sc[C_UREFNZ]; cs[CS_VREF] = sc[C_VREFNZ];
          ci[CI_STATUS] = si[I_STATUS]; ci[CI_NPASS] = si[I_NPASS];
        }
      }
    END_ITEMS
    }   // finish round
    }   // pass of the active slots
    if (!finishing && !(any_waiting && (solo_perm || !sparse))) continue;   // (a workgroup in a view of a few slots asks for its waiting tickets when their columns are done)
    // (every read of the finished slots' records and of the flags M0 rewrites is done: hand the slots to the queue;
    // every wave's stores of the finishing steps have left it - M0 publishes those steps)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    STAMP(29);
    if (wv == mgr) {
      // nothing to work on but a ticket whose column another workgroup still has in its previous step: ask again in a while
      if (!any_active && !first_iteration) __builtin_amdgcn_s_sleep(64);
      M0();
    }
    __syncthreads();
    STAMP(30);
    first_iteration = false;
  }
#ifdef MCKPP_PS_STAMPS
  if (p.dbg && wv == mgr && lane == 0) {
    for (int i = 0; i < 23; ++i) atomicAdd((unsigned long long *)p.dbg + i, tacc[i]);
    atomicAdd((unsigned long long *)p.dbg + 24, tacc[24]);   // of M4: its forward part
    atomicAdd((unsigned long long *)p.dbg + 25, tacc[25]);   // of M5: its forward part
    for (int i = 26; i < 31; ++i) atomicAdd((unsigned long long *)p.dbg + i, tacc[i]);   // parts of the finish round
    atomicAdd((unsigned long long *)p.dbg + 31, tacc[23]);
  }
#endif
#undef STAMP
#undef FOR_ITEMS
#undef END_ITEMS
}

struct ps_geom { int nw, w, per_cu; };

// Slots per workgroup / waves / workgroups per CU.  A pass of a workgroup is its manager wave's serial phases
// (independent of the number of slots: 0.8-1 k cycles per level) plus its level phases (25-26 k cycles per trip
// of the item loop, the waits at their barriers included) plus finish rounds; 128 VGPRs allow 16 waves per CU.
// The rate is slots in flight / pass time: take the geometry that maximises it under the LDS each workgroup's
// slots need, with no more slots than the CU has columns to work on (fewer items, shorter passes).  The
// constants are fits to the per-phase cycle counts of profiles/r02/stamps.txt (one 16-wave workgroup: nothing
// overlaps its barriers; four 4-wave workgroups: four manager waves share the CU with few level waves).
// <= 21 slots: three manager lanes per slot.
ps_geom ps_choose(int L, int xv, size_t cu_lds_bytes, int cols_per_cu, int *max_slots_per_cu)
{
  auto granules = [&](int w_) { return (ps_lds_bytes(L, w_, xv) + 1279) / 1280 * 1280; };
  ps_geom best = {1, 1, 1};
  double best_rate = 0.0;
  int most = 1;
  for (int per_cu = 1; per_cu <= 4; per_cu *= 2) {
    const int nw = 16 / per_cu, threads = 64 * nw;
    const double serial = per_cu == 1 ? 0.82e3 : per_cu == 2 ? 0.92e3 : 1.03e3;
    const double trip = per_cu == 1 ? 26.e3 : per_cu == 2 ? 26.e3 : 24.75e3;   // incl. the waits at the phases' barriers
    const double other = per_cu == 1 ? 24.e3 : per_cu == 2 ? 8.3e3 : 8.6e3;
    for (int w = 1; w <= 21; ++w) {
      if ((size_t)per_cu * granules(w) > cu_lds_bytes) break;
      if (per_cu * w > most) most = per_cu * w;
      const int later = threads > 64 ? threads - 64 : 64;   // the manager wave takes items in the first trip only
      const int trips = w * L <= threads ? 1 : 1 + (w * L - threads + later - 1) / later;
      const double pass = serial * L + trip * trips + other;
      const int busy = per_cu * w < cols_per_cu ? per_cu * w : cols_per_cu;
      const double rate = busy / pass;
      if (rate > best_rate * 1.0001) { best_rate = rate; best = {nw, w, per_cu}; }
    }
  }
  int need = (best.w * L + 63) / 64;
  if (best.w == 1) ++need;   // a workgroup of one slot: a wave for the manager beside the waves of the level items (the view of one slot)
  if (need < best.nw) best.nw = need;
  if (max_slots_per_cu) *max_slots_per_cu = most;
  return best;
}

// MCKPP_PS=<slots>x<waves>x<workgroups per CU> overrides the choice (experiments)
ps_geom ps_geometry(int L, int xv, int cols_per_cu, int *max_slots_per_cu)
{
  ps_geom g = ps_choose(L, xv, (size_t)160 * 1024, cols_per_cu, max_slots_per_cu);
  if (const char *e = getenv("MCKPP_PS")) {
    int w = 0, nw = 0, b = 0;
    if (sscanf(e, "%dx%dx%d", &w, &nw, &b) == 3 && w >= 1 && w <= 21 && nw >= 1 && nw <= 16 && b >= 1 && b <= 16) {
      g = {nw, w, b};
      if (max_slots_per_cu && b * w > *max_slots_per_cu) *max_slots_per_cu = b * w;
    }
  }
  return g;
}

}  // namespace

// the scratch block covers whatever geometry a launch may choose (the choice depends on the column count)
size_t mckpp_ps_scratch_doubles(int nzp1, int xv, int num_cu)
{
  int most = 1;
  (void)ps_geometry(nzp1 + 2, xv, 1 << 20, &most);
  return (size_t)num_cu * most * 4 * ps_scratch_ld(nzp1);
}

hipError_t mckpp_launch_column_kernel_ps(const mckpp_kparams &p, const mckpp_kparams *dp, int num_cu, hipStream_t stream,
                                         mckpp_launch_info *info)
{
  if (p.ncol <= 0) return hipSuccess;
  const int L = p.nzp1 + 2;
  if (L > 1024) return hipErrorInvalidValue;
  const int xv = p.ext ? (p.LDD ? 2 : 1) : 0;   // kernel variant: default physics / optional / optional with double diffusion
  const ps_geom g = ps_geometry(L, xv, (p.ncol + num_cu - 1) / num_cu, nullptr);
  if (!p.scratch || p.scratch_doubles < (size_t)num_cu * g.per_cu * g.w * 4 * ps_scratch_ld(p.nzp1)) return hipErrorInvalidValue;
  const size_t lds = ps_lds_bytes(L, g.w, xv);
  if (lds > (size_t)160 * 1024) return hipErrorInvalidValue;
  using kern_t = void (*)(const mckpp_kparams *, int, int, int, unsigned, int);
  // the default-physics kernels of BASELINE's shapes (60, 69, 100 levels) with the number of level items as a literal: half the
  // spilled SGPRs (104 -> 55), 6 % fewer vector instructions in the item loops; +3 % at 60 levels, +1.4 % at 100
  // (MCKPP_PS_FIXED_L=0: the general kernels, for A/B runs and tests)
  static const kern_t kerns_63[2] = {k_column_ps<0, 0, 63>, k_column_ps<0, 1, 63>}, kerns_72[2] = {k_column_ps<0, 0, 72>, k_column_ps<0, 1, 72>},
                      kerns_103[2] = {k_column_ps<0, 0, 103>, k_column_ps<0, 1, 103>};
  // ... and the optional-physics kernels (relaxation, flux corrections, advection, ...; not double diffusion) of the same shapes
  static const kern_t kernx_63[2] = {k_column_ps<1, 0, 63>, k_column_ps<1, 1, 63>}, kernx_72[2] = {k_column_ps<1, 0, 72>, k_column_ps<1, 1, 72>},
                      kernx_103[2] = {k_column_ps<1, 0, 103>, k_column_ps<1, 1, 103>};
  static const kern_t kerns[2][3] = {{k_column_ps<0, 0>, k_column_ps<1, 0>, k_column_ps<2, 0>},
                                     {k_column_ps<0, 1>, k_column_ps<1, 1>, k_column_ps<2, 1>}};
  if (p.solver_mode < 0 || p.solver_mode > 1) return hipErrorInvalidValue;
  const bool fixed_l = !(getenv("MCKPP_PS_FIXED_L") && atoi(getenv("MCKPP_PS_FIXED_L")) == 0);   // (read at every launch: tests switch it)
  const kern_t kern = (fixed_l && xv == 0 && L == 63) ? kerns_63[p.solver_mode] : (fixed_l && xv == 0 && L == 72) ? kerns_72[p.solver_mode]
                    : (fixed_l && xv == 0 && L == 103) ? kerns_103[p.solver_mode]
                    : (fixed_l && xv == 1 && L == 63) ? kernx_63[p.solver_mode] : (fixed_l && xv == 1 && L == 72) ? kernx_72[p.solver_mode]
                    : (fixed_l && xv == 1 && L == 103) ? kernx_103[p.solver_mode] : kerns[p.solver_mode][xv];
  const void *fn = reinterpret_cast<const void *>(kern);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  int nblocks = num_cu * g.per_cu;
  const int groups = (p.ncol + g.w - 1) / g.w;
  if (nblocks > groups) nblocks = groups;
  if (nblocks < 1) nblocks = 1;
  if (info) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * g.nw, lds) != hipSuccess) nb = 0;
    *info = {nblocks, 64 * g.nw, nb, lds};
  }
  if (getenv("MCKPP_PS_VERBOSE")) {
    static bool said = false;
    if (!said) {
      said = true;
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * g.nw, lds) != hipSuccess) nb = -1;
      fprintf(stderr, "[mckpp ps] L=%d: %d slots x %d waves x %d workgroups per CU, %zu B of LDS each (slot stride %d doubles = %d mod 32, %d of them padding), %d fit on a CU, %d workgroups\n",
              L, g.w, g.nw, g.per_cu, lds, ps_ss(L, xv, g.w), ps_ss(L, xv, g.w) & 31, ps_ss(L, xv, g.w) - ps_rows(xv) * ps_nl(L), nb, nblocks);
    }
  }
  const unsigned Lmagic = (unsigned)(0x100000000ull / (unsigned long long)L) + 1u;   // it / L == umulhi(it, Lmagic) for it < 2^20
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(64 * g.nw), lds, stream, dp, p.ntime, L, g.w, Lmagic, ps_ss(L, xv, g.w));
  return hipGetLastError();
}
