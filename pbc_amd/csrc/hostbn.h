// hostbn.h -- tiny host-side multi-precision helpers used once per pairing_init to derive
// the per-curve constants the kernels keep in __constant__ memory (R mod q, R^2 mod q,
// -q^-1 mod 2^32, q-2, cofactor words).  The reference derives the same constants with GMP
// in field_init_mont_fp (arith/montfp.c:533-600); the product has no GMP dependency.
// Also: parser for PBC's "key value" .param text (ecc/param.c:100-170).
#pragma once
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

namespace pbc_host {

struct Big {                              // little-endian 32-bit words, normalised (no high zeros)
  std::vector<uint32_t> w;

  void trim() { while (!w.empty() && w.back() == 0) w.pop_back(); }
  bool is_zero() const { return w.empty(); }
  int bits() const {
    if (w.empty()) return 0;
    uint32_t t = w.back();
    int b = 0;
    while (t) { b++; t >>= 1; }
    return (int) (w.size() - 1) * 32 + b;
  }
  int bit(int i) const { return (size_t) (i >> 5) < w.size() ? (w[i >> 5] >> (i & 31)) & 1 : 0; }
  static bool from_dec(Big &r, const std::string &s) {
    r.w.clear();
    if (s.empty()) return false;
    for (char ch : s) {
      if (ch < '0' || ch > '9') return false;
      uint64_t c = (uint64_t) (ch - '0');
      for (auto &x : r.w) { c += (uint64_t) x * 10; x = (uint32_t) c; c >>= 32; }
      if (c) r.w.push_back((uint32_t) c);
    }
    r.trim();
    return true;
  }
  static int cmp(const Big &a, const Big &b) {
    if (a.w.size() != b.w.size()) return a.w.size() < b.w.size() ? -1 : 1;
    for (size_t i = a.w.size(); i-- > 0;)
      if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
    return 0;
  }
  void sub(const Big &b) {                 // this -= b, requires this >= b
    uint64_t bw = 0;
    for (size_t i = 0; i < w.size(); i++) {
      uint64_t d = (uint64_t) w[i] - (i < b.w.size() ? b.w[i] : 0) - bw;
      w[i] = (uint32_t) d;
      bw = (d >> 63) & 1;
    }
    trim();
  }
  void shl1() {
    uint32_t c = 0;
    for (auto &x : w) { uint32_t n = x >> 31; x = (x << 1) | c; c = n; }
    if (c) w.push_back(c);
  }
  void sub_small(uint32_t v) { Big b; if (v) b.w.push_back(v); sub(b); }
  // 2^k mod m by k modular doublings of 1
  static Big pow2_mod(int k, const Big &m) {
    Big r; r.w.push_back(1);
    for (int i = 0; i < k; i++) { r.shl1(); if (cmp(r, m) >= 0) r.sub(m); }
    return r;
  }
  void add_small(uint32_t v) {
    uint64_t c = v;
    for (auto &x : w) { c += x; x = (uint32_t) c; c >>= 32; if (!c) break; }
    if (c) w.push_back((uint32_t) c);
  }
  static Big add(const Big &a, const Big &b) {
    Big r;
    uint64_t c = 0;
    for (size_t i = 0; i < a.w.size() || i < b.w.size() || c; i++) {
      c += (uint64_t) (i < a.w.size() ? a.w[i] : 0) + (i < b.w.size() ? b.w[i] : 0);
      r.w.push_back((uint32_t) c);
      c >>= 32;
    }
    r.trim();
    return r;
  }
  static Big mul(const Big &a, const Big &b) {
    Big r;
    r.w.assign(a.w.size() + b.w.size() + 1, 0);
    for (size_t i = 0; i < a.w.size(); i++) {
      uint64_t c = 0;
      for (size_t j = 0; j < b.w.size(); j++) {
        c += (uint64_t) a.w[i] * b.w[j] + r.w[i + j];
        r.w[i + j] = (uint32_t) c;
        c >>= 32;
      }
      r.w[i + b.w.size()] += (uint32_t) c;
    }
    r.trim();
    return r;
  }
  // quotient of an exact or inexact division (binary long division); rem returned via *rem
  static Big div(const Big &a, const Big &b, Big *rem = nullptr) {
    Big q, r;
    q.w.assign(a.w.size() + 1, 0);
    for (int i = a.bits() - 1; i >= 0; i--) {
      r.shl1();
      if (a.bit(i)) { if (r.w.empty()) r.w.push_back(1); else r.w[0] |= 1; }
      if (cmp(r, b) >= 0) { r.sub(b); q.w[i >> 5] |= 1u << (i & 31); }
    }
    q.trim();
    if (rem) *rem = r;
    return q;
  }
  void to_words(uint32_t *out, int n) const {
    for (int i = 0; i < n; i++) out[i] = (size_t) i < w.size() ? w[i] : 0;
  }
};

inline uint32_t neg_inv32(uint32_t p0) {   // -p^-1 mod 2^32 (Newton)
  uint32_t x = 1;
  for (int i = 0; i < 6; i++) x *= 2u - p0 * x;
  return 0u - x;
}

// "key value" lines; returns false if the key is absent
// Signed-digit (non-adjacent form) recoding of the Miller loops' multiplier.  The loops compute f_{r-1,P} as the
// reference does (double-and-add over n = r >> 1, then one closing tangent, e.g. ecc/d_param.c:321-422): with NAF digits
// of n the chain has a third instead of half of its positions non-zero, and the function it builds has the same divisor
// (r - 1)(P) - ((r - 1)P) - (r - 2)(O) up to vertical lines, which the final exponentiation removes -- for every point of
// the curve, not only those of order r.  Digit i of n is stored at bit i + 1 of plus[] / minus[] (the position the
// binary loops test); returns the loop's bit count (index of the leading digit + 1), or 0 when it does not fit `words`.
inline int naf_of_half(const Big &r, uint32_t *plus, uint32_t *minus, int words) {
  for (int i = 0; i < words; i++) plus[i] = minus[i] = 0;
  std::vector<uint32_t> n(r.w);
  n.push_back(0);
  // n = r >> 1
  for (size_t i = 0; i + 1 < n.size(); i++) n[i] = (n[i] >> 1) | (n[i + 1] << 31);
  int pos = 1, top = 0;
  auto is_zero = [&] { for (uint32_t x : n) if (x) return false; return true; };
  while (!is_zero()) {
    if (n[0] & 1) {
      if ((size_t) (pos >> 5) >= (size_t) words) return 0;
      if ((n[0] & 3) == 1) {               // digit +1: n -= 1
        plus[pos >> 5] |= 1u << (pos & 31);
        n[0] -= 1;
      } else {                             // digit -1: n += 1
        minus[pos >> 5] |= 1u << (pos & 31);
        for (size_t i = 0; i < n.size(); i++) if (++n[i]) break;
      }
      top = pos;
    }
    for (size_t i = 0; i + 1 < n.size(); i++) n[i] = (n[i] >> 1) | (n[i + 1] << 31);
    n.back() >>= 1;
    pos++;
  }
  return top + 1;
}

// Subtraction constants of the limb-form kernels (pairing_al.cuh, pairing_d.cuh): a - b is formed as a + (K - b) limb by
// limb with K = c q written in "borrowed" 29-bit limbs -- limb_0 + D 2^29, limb_i + D 2^29 - D, limb_top - D -- so that
// every limb of K dominates the corresponding limb of b.  The kernels' bound trackers (host mirror) prove for every call
// site that  (c - B) 2^(minbits - 1 - 29 (L - 1)) >= D + 1  for the subtrahend's value bound B (in units of q): with
// q >= 2^(minbits - 1) that puts K's top limb above any top limb b can have.  This routine is the other half, run at
// init for the actual q: q has at least `minbits` bits, the limbs are built as the kernels expect, sum to c q, fit 32
// bits and dominate D (2^29 - 1) below the top.  Returns the number of violations (0 = the limb-form path may be used).
inline int ksub_build(const Big &q, int L, int minbits, uint32_t c, uint32_t D, uint32_t *k, int W = 29) {   // W: bits of a limb (28 on the 33-word fields)
  int bad = 0;
  if (q.bits() < minbits || q.bits() > W * L) bad++;
  Big cb;
  cb.w.push_back(c);
  const Big v = Big::mul(q, cb);
  if (v.bits() > W * (L - 1) + 32) bad++;
  for (int i = 0; i < L; i++) {
    uint64_t x = 0;
    for (int b = 0; b < (i < L - 1 ? W : 32); b++) x |= (uint64_t) v.bit(W * i + b) << b;
    x += (i < L - 1 ? (uint64_t) D << W : 0);
    if (i > 0) {
      if (x < D) bad++;
      x -= D;
    }
    if (x >> 32) bad++;
    if (i < L - 1 && x < (uint64_t) D * ((1u << W) - 1)) bad++;
    if (i == L - 1 && x < (uint64_t) D + 1) bad++;
    k[i] = (uint32_t) x;
  }
  Big sum;                                 // sum_i k_i 2^(W i) == c q
  for (int i = L - 1; i >= 0; i--) {
    for (int b = 0; b < W; b++) sum.shl1();
    Big t;
    t.w.push_back(k[i]);
    t.trim();
    sum = Big::add(sum, t);
  }
  if (Big::cmp(sum, v) != 0) bad++;
  return bad;
}

inline bool param_lookup(const char *txt, size_t len, const char *key, std::string &val) {
  size_t klen = strlen(key), i = 0;
  while (i < len) {
    size_t ls = i;
    while (i < len && txt[i] != '\n') i++;
    size_t le = i;
    if (i < len) i++;
    while (ls < le && (txt[ls] == ' ' || txt[ls] == '\t')) ls++;
    if (le - ls > klen && !memcmp(txt + ls, key, klen) && (txt[ls + klen] == ' ' || txt[ls + klen] == '\t')) {
      size_t vs = ls + klen;
      while (vs < le && (txt[vs] == ' ' || txt[vs] == '\t')) vs++;
      size_t ve = le;
      while (ve > vs && (txt[ve - 1] == ' ' || txt[ve - 1] == '\r' || txt[ve - 1] == '\t')) ve--;
      val.assign(txt + vs, ve - vs);
      return true;
    }
  }
  return false;
}
inline bool param_big(const char *txt, size_t len, const char *key, Big &out) {
  std::string v;
  return param_lookup(txt, len, key, v) && Big::from_dec(out, v);
}
inline bool param_int(const char *txt, size_t len, const char *key, int &out) {
  std::string v;
  if (!param_lookup(txt, len, key, v)) return false;
  out = atoi(v.c_str());
  return true;
}

}  // namespace pbc_host
