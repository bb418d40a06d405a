// host_text.h -- PBC's text formats on wire-format records: what element_snprint /
// element_printf("%B") print, what element_set_str reads, what pbc_param_out_str writes.  Host-side string handling,
// no GPU involved: the formats are defined by the reference's per-field routines
//   F_q / Z_r      decimal integer                      fp_snprint / fp_set_str          arith/montfp.c:177-195
//   quadratic      "[x, y]"                             fq_snprint / fq_set_str          arith/fieldquadratic.c:105-157
//   polymod        "[c0, c1, ..., c(n-1)]"              polymod_snprint / _set_str       arith/poly.c:1221-1283
//   curve point    "[x, y]" or "O"                      curve_snprint / curve_set_str    ecc/curve.c:501-578
//   GT             the underlying field element         mulg_snprint                     ecc/pairing.c:187-190
//   parameters     "type t\nkey value\n..."             *_out_str, param_out_mpz         ecc/{a,d,e,f,g}_param.c, ecc/param.c:115-132
// and the numbers go through pbc_mpz_set_str (arith/field.c:725-755: blanks are skipped anywhere, the number ends at the
// first character that is not a digit of the base).  Records are element_to_bytes output: big-endian coordinates of
// len_fq (len_zr) bytes, x before y, polymod coefficients in order.
#pragma once
#include <ctype.h>

#include "hostbn.h"

namespace pbc_host {

inline std::string big_to_dec(const Big &a) {
  if (a.is_zero()) return "0";
  std::vector<uint32_t> w = a.w;
  std::string out;
  while (!w.empty()) {                   // peel nine decimal digits per pass
    uint64_t rem = 0;
    for (size_t i = w.size(); i-- > 0;) {
      const uint64_t cur = (rem << 32) | w[i];
      w[i] = (uint32_t) (cur / 1000000000u);
      rem = cur % 1000000000u;
    }
    while (!w.empty() && w.back() == 0) w.pop_back();
    for (int d = 0; d < 9; d++) {
      out.push_back((char) ('0' + rem % 10));
      rem /= 10;
      if (w.empty() && rem == 0) break;
    }
  }
  while (out.size() > 1 && out.back() == '0') out.pop_back();
  return std::string(out.rbegin(), out.rend());
}
// The integer syntax of element_set_str (what pbc_mpz_set_str, arith/field.c:725-755, accepts): digits of `base`
// (2..36, 0 meaning 10; letters of either case count from ten) with white space allowed anywhere among them; the
// number ends at the first character that is neither.  Returns the characters consumed, white space included (0 for a
// base outside the range).  Digits are gathered into one machine word per group -- as many as keep base^g below 2^32 --
// so the multi-word value takes one multiply-add pass per group rather than per digit.
struct DigitTable {
  int8_t v[256];
  constexpr DigitTable() : v() {
    for (int c = 0; c < 256; c++) v[c] = -1;
    for (int d = 0; d < 10; d++) v['0' + d] = (int8_t) d;
    for (int d = 0; d < 26; d++) { v['A' + d] = (int8_t) (10 + d); v['a' + d] = (int8_t) (10 + d); }
  }
};
inline int big_from_str(Big &z, const char *s, int base) {
  static constexpr DigitTable kDigit;
  z.w.clear();
  const uint32_t radix = base ? (uint32_t) base : 10u;
  if (radix < 2 || radix > 36) return 0;
  auto push = [&z](uint32_t scale, uint32_t add) {      // z = z * scale + add
    uint64_t carry = add;
    for (auto &x : z.w) { carry += (uint64_t) x * scale; x = (uint32_t) carry; carry >>= 32; }
    if (carry) z.w.push_back((uint32_t) carry);
  };
  const char *p = s;
  uint32_t group = 0, scale = 1;
  for (; *p; p++) {
    const unsigned char c = (unsigned char) *p;
    if (isspace(c)) continue;
    const int d = kDigit.v[c];
    if (d < 0 || (uint32_t) d >= radix) break;
    group = group * radix + (uint32_t) d;
    scale *= radix;
    if (scale > 0xffffffffu / radix) { push(scale, group); group = 0; scale = 1; }
  }
  if (scale > 1) push(scale, group);
  z.trim();
  return (int) (p - s);
}
inline Big big_from_be(const uint8_t *p, int n) {
  Big r;
  r.w.assign((size_t) (n + 3) / 4, 0);
  for (int i = 0; i < n; i++) r.w[(size_t) (n - 1 - i) / 4] |= (uint32_t) p[i] << (8 * ((n - 1 - i) % 4));
  r.trim();
  return r;
}
inline void big_to_be(uint8_t *p, int n, const Big &a) {
  for (int i = 0; i < n; i++) {
    const size_t k = (size_t) (n - 1 - i);
    p[i] = k / 4 < a.w.size() ? (uint8_t) (a.w[k / 4] >> (8 * (k % 4))) : 0;
  }
}
inline Big big_mod(const Big &a, const Big &m) {
  Big r;
  if (Big::cmp(a, m) < 0) return a;
  Big::div(a, m, &r);
  return r;
}

// The shape of an element: a tower over F_q (or Z_r), optionally the two coordinates of a curve point
struct TextShape {
  int len;                               // bytes of a base-field coordinate
  int quad;                              // the coordinate field has a quadratic layer on top / below (see order)
  int poly;                              // polymod degree (0: none)
  bool quad_outer;                       // quadratic extension OF the polymod (types d / g GT) rather than polymod of quadratics (type f GT)
  bool curve;
  int coord_bytes() const { return len * (quad ? 2 : 1) * (poly ? poly : 1); }
};

// field element at p -> text
// (coordinates are reduced mod `modulus` first, as fp_from_bytes does)
inline void text_field(std::string &out, const TextShape &S, const uint8_t *p, const Big &modulus) {
  const bool has_quad = S.quad != 0, has_poly = S.poly != 0;
  auto base = [&](const uint8_t *q) { out += big_to_dec(big_mod(big_from_be(q, S.len), modulus)); };
  auto quad_of = [&](const uint8_t *q, auto &&inner, int inner_bytes) {
    out += "[";
    inner(q);
    out += ", ";
    inner(q + inner_bytes);
    out += "]";
  };
  auto poly_of = [&](const uint8_t *q, auto &&inner, int inner_bytes) {
    out += "[";
    for (int i = 0; i < S.poly; i++) {
      if (i) out += ", ";
      inner(q + (size_t) i * inner_bytes);
    }
    out += "]";
  };
  if (!has_quad && !has_poly) { base(p); return; }
  if (has_quad && !has_poly) { quad_of(p, base, S.len); return; }
  if (!has_quad && has_poly) { poly_of(p, base, S.len); return; }
  if (S.quad_outer) {
    auto inner = [&](const uint8_t *q) { poly_of(q, base, S.len); };
    quad_of(p, inner, S.len * S.poly);
  } else {
    auto inner = [&](const uint8_t *q) { quad_of(q, base, S.len); };
    poly_of(p, inner, 2 * S.len);
  }
}

// text -> field element at p (reduced mod `modulus`); characters consumed, 0 on a syntax error (the element is zeroed,
// as element_set0 at the head of the reference's routines)
inline int parse_field(const TextShape &S, uint8_t *p, const char *s, int base, const Big &modulus) {
  const int total = S.coord_bytes();
  memset(p, 0, (size_t) total);
  auto num = [&](uint8_t *q, const char *c) -> int {
    Big z;
    const int used = big_from_str(z, c, base);
    big_to_be(q, S.len, big_mod(z, modulus));
    return used;
  };
  auto skip = [](const char *&c) { while (*c && isspace((unsigned char) *c)) c++; };
  // generic bracketed list of `count` items of `item_bytes` each, parsed by `item`
  auto list = [&](uint8_t *q, const char *c0, int count, int item_bytes, bool space_before_comma, auto &&item) -> int {
    const char *c = c0;
    skip(c);
    if (*c++ != '[') return 0;
    for (int i = 0; i < count; i++) {
      const int used = item(q + (size_t) i * item_bytes, c);
      if (used < 0) return 0;
      c += used;
      if (space_before_comma || i < count - 1) skip(c);       // fq_set_str skips blanks only before the comma, polymod before both
      if (i < count - 1 && *c++ != ',') return 0;
    }
    if (*c++ != ']') return 0;
    return (int) (c - c0);
  };
  const bool has_quad = S.quad != 0, has_poly = S.poly != 0;
  if (!has_quad && !has_poly) return num(p, s);
  auto quad_base = [&](uint8_t *q, const char *c) { return list(q, c, 2, S.len, false, num); };
  auto poly_base = [&](uint8_t *q, const char *c) { return list(q, c, S.poly, S.len, true, num); };
  int used;
  if (has_quad && !has_poly) used = quad_base(p, s);
  else if (!has_quad && has_poly) used = poly_base(p, s);
  else if (S.quad_outer) {
    auto inner = [&](uint8_t *q, const char *c) { const int u = poly_base(q, c); return u ? u : -1; };
    used = list(p, s, 2, S.len * S.poly, false, inner);
  } else {
    auto inner = [&](uint8_t *q, const char *c) { const int u = quad_base(q, c); return u ? u : -1; };
    used = list(p, s, S.poly, 2 * S.len, true, inner);
  }
  if (!used) memset(p, 0, (size_t) total);
  return used;
}

}  // namespace pbc_host
