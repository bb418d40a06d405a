// pairing_ew.cuh -- type e (k = 1: G1 = G2 = E(F_q)[r], GT = F_q; e.param: a 1020-bit q) on the limb-per-lane routines of
// pairing_aw.cuh: ONE PAIRING PER WORKGROUP of four wavefronts (or per wavefront), the small-batch path of a family whose lane
// kernel needs 35 ms for a pairing however small the batch (round 6).  An F_q element is one register (lane j holds limb j: 38
// limbs of 28 bits on the 33-word fields), products run across the lanes, up to four independent ones per round (AW::mul4 /
// round_nw), sums and differences are single instructions with AG<N>'s borrowed constants.
// The ALGORITHM is pairing_e.cuh's (e_miller_lane / e_double_step / e_add_step / e_prod_pairing_lane, which restate e_pairing,
// ecc/e_param.c:472-483, with Jacobian steps, the verticals kept, numerator and denominator apart, one inversion and one
// (q - 1) / r power): the same formulas regrouped into rounds, so the canonical bytes are the same on every input.
//   doubling step, six rounds (s = x_R, a small integer, as an element; ZZ x1 and s ZZ come from the previous step's verticals):
//     1  X^2, Y^2, Z^4 = ZZ^2, Z3 = 2Y Z                 2  n^2, d^2, a Z^4, W = Z3 ZZ            -> M = a Z^4 + 3X^2
//     3  l(S1) = W y1 + M (X - ZZ x1), l(S2) = W y2 + M (X - s ZZ) (two-product sums), S1 = Y^2 2X, Y^4      -> l - 2Y^2
//     4  M^2, ZZ' = Z3^2, n^2 l(S1), d^2 l(S2)           -> X3 = M^2 - 4 S1, Wd = 2 S1 - X3
//     5  M Wd, s ZZ', ZZ' x1                              -> Y3 = M Wd - 8Y^4, v(S2) = s ZZ' - X3, v(S1) = ZZ' x1 - X3
//     6  n v(S2), d v(S1)
//   bounds: every product's operands are products' outputs or normalised sums (limbs ~2^28, one operand may be a sum of two); the
//   subtrahends' values stay below the constant that dominates them (noted per line: B = value in units of q), with q filling
//   twelve bits of its top limb and 44 bits of the radix free (host_params.h ag_aux_build checks both for the object's q).
// Host mirror (tests/hostsim): the same source lane by lane with an element = all its limbs, under the bound tracker of AG<N>'s
// mirror -- which is where the bounds noted below are checked for every operation of a pairing; tests/test_gpu_agwave.py compares
// the device with the lane kernel, the reference's vectors and the C restatement.
#pragma once
#include "pairing_aw.cuh"
#include "pairing_e.cuh"

namespace pbc {

template <int N, int NW>
struct EW : AW<N, NW, AG<N>> {
  typedef AW<N, NW, AG<N>> B;
  typedef typename B::W W;
  typedef typename B::el el;
  typedef AG<N> A;
  static constexpr int L = B::L;
  enum { K2 = B::K2, K4 = B::K4, K8 = B::K8, K12 = B::K12, K16 = B::K16 };
  using B::add; using B::norm; using B::subk; using B::mul; using B::sqr; using B::mul2; using B::mul4; using B::sop2x2; using B::sync;
  using B::load_uniform; using B::invert_lane0; using B::init;
#ifndef PBC_HOSTSIM
  using B::lane; using B::put_slot; using B::get_slot; using B::slot_to_el; using B::el_to_slot;
  static PBC_DEV W zero_w() { return 0u; }
#else
  static W zero_w() { W z; for (int i = 0; i < L; i++) z.l[i] = 0; A::hs_set(z, 0.0, 0.0); return z; }
#endif

  struct est { W n, d, X, Y, Z, ZZ, sZZ, ZZx1; };
  W cA, cS, cY2, x1, y1, xP, yP, oneR;         // a, s = x_R, y_R (uniform); S1 = Q + R; P; R mod q
  bool ok;

  // four sums of two products in one round (NW = 1: two instruction streams of two)
  PBC_DEV void sop4(W &r0, W &r1, W &r2, W &r3, W a0, W b0, W a1, W b1, W c0, W d0, W c1, W d1, W e0, W f0, W e1, W f1, W g0, W h0, W g1, W h1) {
    if constexpr (NW == 1) {
      sop2x2(r0, r1, a0, b0, a1, b1, c0, d0, c1, d1);
      sop2x2(r2, r3, e0, f0, e1, f1, g0, h0, g1, h1);
    } else {
      W r[4];
      const W x0[4] = {a0, c0, e0, g0}, y0[4] = {b0, d0, f0, h0}, x1_[4] = {a1, c1, e1, g1}, y1_[4] = {b1, d1, f1, h1};
      this->template round_nw<2>(r, x0, y0, x1_, y1_, 4);
      r0 = r[0]; r1 = r[1]; r2 = r[2]; r3 = r[3];
    }
  }
  // P, Q -> the wave's registers; Q + R affine (element_add(QR, Q, p->R), e_param.c:479) in word form.  Every lane of every wave
  // runs the same word-form code on the same data (no call under a partial EXEC mask: tools/gpu_faults.md); thread 0 hands the
  // limbs over through the LDS slots.
  PBC_DEV void setup(const uint8_t *g1, const uint8_t *g2) {
    const int NB = (int) fpk<N>().fbytes;
    fp<N> xp, yp, xq, yq, l, t, x3, y3;
    const fp<N> rx = ek<N>(c_e.Rx), ry = ek<N>(c_e.Ry);
    fp_load_be<N>(xp, g1);
    fp_load_be<N>(yp, g1 + NB);
    fp_load_be<N>(xq, g2);
    fp_load_be<N>(yq, g2 + NB);
    ok = (int) e_on_curve<N>(xp, yp) & (int) e_on_curve<N>(xq, yq);
    fp_sub<N>(t, rx, xq);
    fp_inv<N>(t, t);
    fp_sub<N>(l, ry, yq);
    fp_mul<N>(l, l, t);
    fp_sqr<N>(x3, l);
    fp_sub<N>(x3, x3, xq);
    fp_sub<N>(x3, x3, rx);
    fp_sub<N>(t, xq, x3);
    fp_mul<N>(y3, t, l);
    fp_sub<N>(y3, y3, yq);
    el e[4];
    A::to_el(e[0], xp);
    A::to_el(e[1], yp);
    A::to_el(e[2], x3);
    A::to_el(e[3], y3);
#ifdef PBC_HOSTSIM
    xP = e[0]; yP = e[1]; x1 = e[2]; y1 = e[3];
#else
    sync();
    if (threadIdx.x == 0)
      for (int i = 0; i < 4; i++) el_to_slot(e[i], i);
    sync();
    xP = get_slot(0); yP = get_slot(1); x1 = get_slot(2); y1 = get_slot(3);
    sync();
#endif
    cA = load_uniform(c_e.A);
    cS = load_uniform(c_e.Rx);
    cY2 = load_uniform(c_e.Ry);
    oneR = load_uniform(fpk<N>().one);
  }
  PBC_DEV void double_step(est &s) {
    W XX, YY, Z4, Z3, n2, d2, AZ4, Wv, l1, l2, S1, Y4, MM, ZZn, nl, dl, MWd, sZZn, ZZnx1, unused;
    mul4(XX, YY, Z4, Z3, s.X, s.X, s.Y, s.Y, s.ZZ, s.ZZ, B::template shl<1>(s.Y), s.Z, 4);
    mul4(n2, d2, AZ4, Wv, s.n, s.n, s.d, s.d, cA, Z4, Z3, s.ZZ, 4);
    const W M = norm(add(add(B::template shl<1>(XX), XX), AZ4));                       // B 4
    const W g1 = norm(subk(s.X, s.ZZx1, K2)), g2 = norm(subk(s.X, s.sZZ, K2));           // X - ZZ x1, X - s ZZ (B <= 9 + 2)
    const W X2 = B::template shl<1>(s.X), z = zero_w();
    sop4(l1, l2, S1, Y4, Wv, y1, M, g1, Wv, cY2, M, g2, YY, X2, z, z, YY, YY, z, z);
    const W YY2 = B::template shl<1>(YY);
    l1 = norm(subk(l1, YY2, K4));                                                        // - 2Y^2 (B 2 < 4)
    l2 = norm(subk(l2, YY2, K4));
    mul4(MM, ZZn, nl, dl, M, M, Z3, Z3, n2, l1, d2, l2, 4);
    const W X3 = norm(subk(MM, B::template shl<2>(S1), K8));                             // B 9
    const W Wd = norm(subk(B::template shl<1>(S1), X3, K12));                            // 2 S1 - X3 (X3: B 9 < 12)
    mul4(MWd, sZZn, ZZnx1, unused, M, Wd, ZZn, cS, ZZn, x1, ZZn, ZZn, 3);
    const W Y48 = norm(B::template shl<3>(Y4));                                          // B 8
    s.Y = norm(subk(MWd, Y48, K12));                                                     // B 13
    const W v2 = norm(subk(sZZn, X3, K12)), v1 = norm(subk(ZZnx1, X3, K12));
    mul2(s.n, s.d, nl, v2, dl, v1);
    s.X = X3; s.Z = Z3; s.ZZ = ZZn; s.sZZ = sZZn; s.ZZx1 = ZZnx1;
  }
  // V <- V + P; chord scaled by Z3 = Z H: l(S) = (ys - yP) Z3 - R' (xs - xP), H = xP ZZ - X, R' = yP Z^3 - Y (e_add_step)
  PBC_DEV void add_step(est &s, const W &dy1, const W &dx1, const W &dy2, const W &dx2) {
    W a, b, c, Z3, HH, l1, l2, HHH, XHH, RR, ZZn, e, YH, nl, dl, sZZn, ZZnx1, unused;
    mul2(a, b, xP, s.ZZ, s.Z, s.ZZ);
    const W H = norm(subk(a, s.X, K16));                                                 // X: B 9 < 16
    mul4(c, Z3, HH, unused, yP, b, H, s.Z, H, H, H, H, 3);
    const W Rr = norm(subk(c, s.Y, K16));                                                // Y: B 13 < 16
    sop2x2(l1, l2, dy1, Z3, Rr, dx1, dy2, Z3, Rr, dx2);                                  // (ys - yP) Z3 + R' (xP - xs)
    mul4(HHH, XHH, RR, ZZn, HH, H, s.X, HH, Rr, Rr, Z3, Z3, 4);
    W t1 = norm(subk(RR, HHH, K2));
    const W X3 = norm(subk(t1, B::template shl<1>(XHH), K4));                            // B 7
    const W dd = norm(subk(XHH, X3, K16));
    mul4(e, YH, nl, dl, Rr, dd, s.Y, HHH, s.n, l1, s.d, l2, 4);
    s.Y = norm(subk(e, YH, K2));
    mul2(sZZn, ZZnx1, ZZn, cS, ZZn, x1);
    const W v2 = norm(subk(sZZn, X3, K8)), v1 = norm(subk(ZZnx1, X3, K8));               // X3: B 7 < 8
    mul2(s.n, s.d, nl, v2, dl, v1);
    s.X = X3; s.Z = Z3; s.ZZ = ZZn; s.sZZ = sZZn; s.ZZx1 = ZZnx1;
  }
  // numerator and denominator of f_(r, P)(Q + R) / f_(r, P)(R) (e_miller_lane); returns the validity of the inputs
  PBC_DEV bool miller(est &s, const uint8_t *g1, const uint8_t *g2) {
    setup(g1, g2);
    s.n = oneR; s.d = oneR; s.X = xP; s.Y = yP; s.Z = oneR; s.ZZ = oneR; s.sZZ = cS;
    s.ZZx1 = x1;
    const W dy1 = norm(subk(y1, yP, K2)), dx1 = norm(subk(xP, x1, K2)), dy2 = norm(subk(cY2, yP, K2)), dx2 = norm(subk(xP, cS, K2));
    for (int i = c_e.rbits - 2; i >= 0; i--) {
      double_step(s);
      if ((c_e.r[i >> 5] >> (i & 31)) & 1) {
        if (i > 0) {
          add_step(s, dy1, dx1, dy2, dx2);
        } else {                                   // the last addition: V = -P, the chord is the vertical through P
          const W nx = norm(subk(x1, xP, K2)), ndx = norm(subk(cS, xP, K2));
          mul2(s.n, s.d, s.n, nx, s.d, ndx);
        }
      }
    }
    return ok;
  }
  // (n / d)^((q - 1) / r) -> bytes: one inversion (AW::invert_lane0), the sliding window of e_pow_win (odd powers x .. x^15; every
  // wave runs the chain itself: one product at a time, nothing to share)
  PBC_DEV void finish(uint8_t *gt, const W &n, const W &d, bool valid) {
    const W di = invert_lane0(d);
    // (four wavefronts per unit: the chain below is one product at a time -- wave 0 runs it alone and the others leave their SIMDs
    // to other units; no workgroup barrier from here on)
#ifndef PBC_HOSTSIM
    if (NW > 1 && B::wave() != 0) return;
#endif
    const W x = mul(n, di);
    W tab[8];
    {
      const W x2 = sqr(x);
      tab[0] = x;
#pragma unroll
      for (int i = 1; i < 8; i++) tab[i] = mul(tab[i - 1], x2);
    }
    W acc = oneR;
    bool started = false;
    const uint32_t *e = c_e.phik;
    int i = c_e.phikbits - 1;
    while (i >= 0) {
      if (!((e[i >> 5] >> (i & 31)) & 1)) {
        if (started) acc = sqr(acc);
        i--;
        continue;
      }
      int lo = i >= 3 ? i - 3 : 0;
      while (!((e[lo >> 5] >> (lo & 31)) & 1)) lo++;
      int w = 0;
      for (int j = i; j >= lo; j--) w = 2 * w + (int) ((e[j >> 5] >> (j & 31)) & 1);
      W t = tab[0];
#pragma unroll
      for (int k = 1; k < 8; k++) t = (w >> 1) == k ? tab[k] : t;          // (w is wave-uniform: scalar selects, the table stays in registers)
      if (started) {
        for (int j = i; j >= lo; j--) acc = sqr(acc);
        acc = mul(acc, t);
      } else {
        acc = t;
        started = true;
      }
      i = lo - 1;
    }
#ifndef PBC_HOSTSIM
    put_slot(acc, 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
    if (threadIdx.x == 0) {
      el ex;
      fp<N> out;
#ifdef PBC_HOSTSIM
      ex = acc;
#else
      slot_to_el(ex, 0);
#endif
      A::to_words(out, ex);
      if (!valid) fp_set<N>(out, fpk<N>().one);    // GT identity (pairing_apply, include/pbc_pairing.h:123-130)
      fp_store_be<N>(gt, out);
    }
  }
  // element_pairing
  PBC_DEV void pairing_wave(uint8_t *gt, const uint8_t *g1, const uint8_t *g2) {
    est s;
    init();
    const bool valid = miller(s, g1, g2);
    finish(gt, s.n, s.d, valid);
  }
  // element_prod_pairing (generic_prod_pairings, ecc/pairing.c:35-46): a record (n, d, validity) per TERM, then per product the
  // numerators and denominators multiplied, ONE inversion and ONE power
  static constexpr int WREC = B::WREC;
#ifdef PBC_HOSTSIM
  struct wrec { W n, d; bool valid; };
  void miller_record_wave(wrec &r, const uint8_t *g1, const uint8_t *g2) {
    est s;
    init();
    r.valid = miller(s, g1, g2);
    r.n = s.n; r.d = s.d;
  }
  void prod_finish_wave(uint8_t *gt, const wrec *rec, int k) {
    init();
    oneR = load_uniform(fpk<N>().one);
    bool valid = rec[0].valid;
    W n = rec[0].n, d = rec[0].d;
    for (int t = 1; t < k; t++) {
      valid = valid && rec[t].valid;
      mul2(n, d, n, rec[t].n, d, rec[t].d);
    }
    finish(gt, n, d, valid);
  }
#else
  PBC_DEV void miller_record_wave(uint32_t *rec, const uint8_t *g1, const uint8_t *g2) {
    est s;
    init();
    const bool valid = miller(s, g1, g2);
    const int j = lane();
    if (NW == 1 || B::wave() == 0) {
      if (j < L) { rec[j] = s.n; rec[L + j] = s.d; }
      if (threadIdx.x == 0) rec[2 * L] = valid ? 1u : 0u;
    }
  }
  PBC_DEV void prod_finish_wave(uint8_t *gt, const uint32_t *rec, int k) {
    init();
    const int j = lane();
    oneR = load_uniform(fpk<N>().one);
    uint32_t okw = rec[2 * L];
    W n = j < L ? rec[j] : 0u, d = j < L ? rec[L + j] : 0u;
    for (int t = 1; t < k; t++) {
      const uint32_t *r = rec + (size_t) t * WREC;
      okw &= r[2 * L];
      const W rn = j < L ? r[j] : 0u, rd = j < L ? r[L + j] : 0u;
      mul2(n, d, n, rn, d, rd);
    }
    finish(gt, n, d, okw != 0);
  }
#endif
};

}  // namespace pbc
