// sopvm_probe.hip -- PROTOTYPE (round 6, VERDICT r5 "missing" 1): can ONE d159 pairing run below the CPU's 1.37 ms?
//
// The throughput kernels run a pairing as one lane's serial instruction stream: 2.0 M vector instructions, 3.9 ms through
// the hooks whatever the batch size.  Every tower operation of pairing_d.cuh, though, is a set of INDEPENDENT lazily reduced
// sums of F_q products (sop_limbs): an F_q^6 square is 3 + 2 + 3 sums in three dependent levels for the x part and 2 + 3
// in two levels for the y part, the point arithmetic on E(F_q) another handful per level.  This probe measures the
// alternative shape without building the pairing: one pairing per WAVEFRONT, every element a 6-limb slot in LDS, a step =
// a sequence of LEVELS, and in a level lane l computes ONE sum  out[l] = sum_t x[l][t] y[l][t] / R  (operands gathered from
// the slot file by per-lane index tables, T padded to the level's maximum with a zero slot) and writes its slot.  The
// program below has the shape of one Miller doubling step of d159 (f <- f^2, then f <- f * line, the point doubling and
// the line's products riding along in other lanes): levels of 1, 4, 8, 1, 4, 8 terms.  The arithmetic is the library's
// (fp.cuh sop_limbs on f.param's / d159's 6-limb field), so the time per level is what the real kernel would pay.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pbc_amd/csrc tools/sopvm_probe.hip -o /tmp/sopvm_probe && /tmp/sopvm_probe
//
// Output: microseconds per Miller-step-shaped program for 1 ... 4096 concurrent wavefronts, the per-level split, and a
// check of one level against 128-bit host arithmetic.  Extrapolation: profiles/r06_notes.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "fp.cuh"

using namespace pbc;
constexpr int N = 5, L = 6, SLOTS = 96;          // 96 six-limb slots per wavefront: f (6), temporaries, the point, constants
constexpr int TMAX = 8, LEVELS = 6;

struct Level { int T; uint8_t x[64][TMAX], y[64][TMAX], out[64]; };
struct Program { Level lv[LEVELS]; };

template <int T>
__device__ __forceinline__ void run_level(uint32_t *slots, const Level &lv) {
  const int lane = threadIdx.x;
  fl<N> x[T], y[T], r;
#pragma unroll
  for (int t = 0; t < T; t++) {
    const uint32_t *px = slots + lv.x[lane][t] * L, *py = slots + lv.y[lane][t] * L;
#pragma unroll
    for (int i = 0; i < L; i++) { x[t].l[i] = px[i]; y[t].l[i] = py[i]; }
  }
  sop_limbs<N, T>(r, x, y);
  uint32_t *po = slots + lv.out[lane] * L;
#pragma unroll
  for (int i = 0; i < L; i++) po[i] = r.l[i];
}

// one wavefront per workgroup; `steps` repetitions of the program (158 doubling steps make a d159 Miller loop)
template <bool PROG_IN_LDS>
__global__ void __launch_bounds__(64) sopvm_kernel(uint32_t *out, const uint32_t *init, const Program *gprog, int steps, long long *cycles, KArgs<N> ka) {
  __shared__ uint32_t slots[SLOTS * L];
  __shared__ Program lprog;                      // (PROG_IN_LDS: the index tables next to the slot file -- no global load per level)
  for (int i = threadIdx.x; i < SLOTS * L; i += 64) slots[i] = init[i];
  if (PROG_IN_LDS)
    for (int i = threadIdx.x; i < (int) (sizeof(Program) / 4); i += 64) reinterpret_cast<uint32_t *>(&lprog)[i] = reinterpret_cast<const uint32_t *>(gprog)[i];
  const Program *prog = PROG_IN_LDS ? &lprog : gprog;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter(), per[LEVELS] = {0};
  for (int s = 0; s < steps; s++) {
#pragma nounroll
    for (int v = 0; v < LEVELS; v++) {
      const Level &lv = prog->lv[v];
      const long long a = __builtin_readcyclecounter();
      switch (lv.T) {                            // wave-uniform
        case 1: run_level<1>(slots, lv); break;
        case 2: run_level<2>(slots, lv); break;
        case 4: run_level<4>(slots, lv); break;
        default: run_level<8>(slots, lv); break;
      }
      __builtin_amdgcn_wave_barrier();
      per[v] += __builtin_readcyclecounter() - a;
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    cycles[0] = t1 - t0;
    for (int v = 0; v < LEVELS; v++) cycles[1 + v] = per[v];
  }
  for (int i = threadIdx.x; i < SLOTS * L; i += 64) out[(size_t) blockIdx.x * SLOTS * L + i] = slots[i];
}

// ---- host side: the 158-bit field of f.param (same width as d159: six 29-bit limbs) ----
typedef unsigned __int128 u128;
static const char *Q_DEC = "205523667896953300194896352429254920972540065223";     // f.param q
struct Big5 { uint32_t w[6]; };
static void dec_to_words(const char *s, uint32_t *w, int n) {
  memset(w, 0, 4 * n);
  for (; *s; s++) {
    uint64_t c = (uint64_t) (*s - '0');
    for (int i = 0; i < n; i++) { c += (uint64_t) w[i] * 10; w[i] = (uint32_t) c; c >>= 32; }
  }
}
int main() {
  uint32_t qw[6];
  dec_to_words(Q_DEC, qw, 6);
  KArgs<N> K;
  memset(&K, 0, sizeof K);
  for (int i = 0; i < N; i++) K.fp.p[i] = qw[i];
  for (int i = 0; i < L; i++) {
    uint32_t v = 0;
    for (int b = 0; b < 29; b++) { const int bit = 29 * i + b; v |= ((qw[bit >> 5] >> (bit & 31)) & 1u) << b; }
    K.fp.p29[i] = v;
  }
  uint32_t inv = 1;
  for (int i = 0; i < 5; i++) inv *= 2 - qw[0] * inv;          // q^-1 mod 2^32
  K.fp.ninv29 = (0u - inv) & ((1u << 29) - 1);
  K.fp.pbits = 158; K.fp.fbytes = 20;
  // slot file: random limbs below 2^29 (slot 0: zero -- the padding operand)
  std::vector<uint32_t> init(SLOTS * L);
  srand(7);
  for (auto &v : init) v = ((uint32_t) rand() << 14 ^ (uint32_t) rand()) & ((1u << 29) - 1);
  for (int i = 0; i < L; i++) init[i] = 0;
  for (int s = 0; s < SLOTS; s++) init[s * L + L - 1] &= (1u << 12) - 1;     // values below 2^157 < q: valid operands of a lazy sum
  // the program: a Miller doubling step's shape.  Lanes 0-13 work (f^2: 3 + 2 + 3 | 2 + 3 sums; the point doubling and the
  // line: up to 6 sums per level), the others compute padding.  Terms per level: 1, 4, 8 (f^2), 1, 4, 8 (f * line).
  Program P;
  const int Ts[LEVELS] = {1, 4, 8, 1, 4, 8};
  for (int v = 0; v < LEVELS; v++) {
    P.lv[v].T = Ts[v];
    for (int l = 0; l < 64; l++) {
      for (int t = 0; t < TMAX; t++) { P.lv[v].x[l][t] = (uint8_t) (l < 14 ? 1 + (l * 7 + t * 3 + v) % 40 : 0); P.lv[v].y[l][t] = (uint8_t) (l < 14 ? 1 + (l * 5 + t * 11 + 2 * v) % 40 : 0); }
      P.lv[v].out[l] = (uint8_t) (l < 14 ? 41 + (l + 14 * (v & 1)) % 28 : 70 + (l % 26));    // outputs never alias this level's operands
    }
  }
  uint32_t *d_init, *d_out;
  Program *d_prog;
  long long *d_cyc;
  const int maxw = 4096;
  hipMalloc(&d_init, init.size() * 4); hipMalloc(&d_out, (size_t) maxw * SLOTS * L * 4); hipMalloc(&d_prog, sizeof P); hipMalloc(&d_cyc, 8 * 8);
  hipMemcpy(d_init, init.data(), init.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_prog, &P, sizeof P, hipMemcpyHostToDevice);
  // ---- check: one level of T = 8 against host arithmetic (value mod q) ----
  {
    hipLaunchKernelGGL(sopvm_kernel<false>, dim3(1), dim3(64), 0, 0, d_out, d_init, d_prog, 0, d_cyc, K);   // steps = 0: copies the slots
    Program P1 = P;
    for (int v = 0; v < LEVELS; v++) P1.lv[v] = P.lv[2];
    Program *d_p1; hipMalloc(&d_p1, sizeof P1); hipMemcpy(d_p1, &P1, sizeof P1, hipMemcpyHostToDevice);
    // run level 2 once: steps = 1 runs it six times on the same operands (outputs do not alias operands), same result
    hipLaunchKernelGGL(sopvm_kernel<false>, dim3(1), dim3(64), 0, 0, d_out, d_init, d_p1, 1, d_cyc, K);
    std::vector<uint32_t> got(SLOTS * L);
    hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost);
    // host: value(slot) = sum l_i 2^(29 i); want sum x y / 2^174 mod q == value(out) mod q  <=>  value(out) 2^174 == sum x y (mod q)
    auto val_mod_q = [&](const uint32_t *l, int shift_limbs, u128 *hi_unused) {
      (void) hi_unused;
      // big arithmetic through repeated (mod q) Horner steps on 29-bit digits, all in 6-word integers
      uint32_t acc[7] = {0};
      auto mul2_29_add = [&](uint32_t d) {
        // acc = (acc * 2^29 + d) mod q
        uint64_t c = d;
        uint32_t t[8] = {0};
        for (int i = 0; i < 6; i++) { uint64_t v = ((uint64_t) acc[i] << 29) + c; t[i] = (uint32_t) v; c = v >> 32; }
        t[6] = (uint32_t) c;
        // reduce t (< 2^(192+29)) mod q by shift-subtract
        for (int sh = 63; sh >= 0; sh--) {
          uint32_t qs[8] = {0};
          for (int i = 0; i < 6; i++) { const int bit = 32 * i + sh; qs[bit >> 5] |= qw[i] << (bit & 31); if (bit & 31) qs[(bit >> 5) + 1] |= qw[i] >> (32 - (bit & 31)); }
          bool ge = true;
          for (int i = 7; i >= 0; i--) if (t[i] != qs[i]) { ge = t[i] > qs[i]; break; }
          if (ge) { uint64_t b = 0; for (int i = 0; i < 8; i++) { uint64_t v = (uint64_t) t[i] - qs[i] - b; t[i] = (uint32_t) v; b = (v >> 63) & 1; } }
        }
        for (int i = 0; i < 6; i++) acc[i] = t[i];
      };
      for (int i = L - 1; i >= 0; i--) mul2_29_add(l[i]);
      for (int i = 0; i < shift_limbs; i++) mul2_29_add(0);
      Big5 r; for (int i = 0; i < 6; i++) r.w[i] = acc[i];
      return r;
    };
    auto mulmod = [&](const Big5 &a, const Big5 &b) {
      // schoolbook by bits: r = a * b mod q
      uint32_t r[7] = {0};
      for (int bit = 191; bit >= 0; bit--) {
        uint64_t c = 0;
        for (int i = 0; i < 7; i++) { uint64_t v = ((uint64_t) r[i] << 1) | c; r[i] = (uint32_t) v; c = v >> 32; }
        if ((b.w[bit >> 5] >> (bit & 31)) & 1) { uint64_t cc = 0; for (int i = 0; i < 7; i++) { uint64_t v = (uint64_t) r[i] + (i < 6 ? a.w[i] : 0) + cc; r[i] = (uint32_t) v; cc = v >> 32; } }
        for (int rep = 0; rep < 2; rep++) {
          bool ge = true;
          for (int i = 6; i >= 0; i--) { const uint32_t qi = i < 6 ? qw[i] : 0; if (r[i] != qi) { ge = r[i] > qi; break; } }
          if (ge) { uint64_t bw = 0; for (int i = 0; i < 7; i++) { uint64_t v = (uint64_t) r[i] - (i < 6 ? qw[i] : 0) - bw; r[i] = (uint32_t) v; bw = (v >> 63) & 1; } }
        }
      }
      Big5 o; for (int i = 0; i < 6; i++) o.w[i] = r[i];
      return o;
    };
    int bad = 0;
    for (int l = 0; l < 14; l++) {
      Big5 sum = {{0}};
      for (int t = 0; t < 8; t++) {
        Big5 a = val_mod_q(&init[P.lv[2].x[l][t] * L], 0, nullptr), b = val_mod_q(&init[P.lv[2].y[l][t] * L], 0, nullptr), p = mulmod(a, b);
        uint64_t c = 0;
        for (int i = 0; i < 6; i++) { uint64_t v = (uint64_t) sum.w[i] + p.w[i] + c; sum.w[i] = (uint32_t) v; c = v >> 32; }
        bool ge = true;
        for (int i = 5; i >= 0; i--) if (sum.w[i] != qw[i]) { ge = sum.w[i] > qw[i]; break; }
        if (ge) { uint64_t bw = 0; for (int i = 0; i < 6; i++) { uint64_t v = (uint64_t) sum.w[i] - qw[i] - bw; sum.w[i] = (uint32_t) v; bw = (v >> 63) & 1; } }
      }
      Big5 lhs = val_mod_q(&got[P.lv[2].out[l] * L], L, nullptr);     // value(out) 2^174 mod q
      if (memcmp(lhs.w, sum.w, 24)) bad++;
    }
    printf("check: a level of 8-term sums on 14 lanes against host arithmetic: %s\n", bad ? "MISMATCH" : "ok");
    hipFree(d_p1);
  }
  // ---- timing ----
  const int steps = 158;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int in_lds = 0; in_lds < 2; in_lds++)
  for (int waves : {1, 16, 256, 1024, 4096}) {
    auto launch = [&]() {
      if (in_lds) hipLaunchKernelGGL(sopvm_kernel<true>, dim3(waves), dim3(64), 0, 0, d_out, d_init, d_prog, steps, d_cyc, K);
      else hipLaunchKernelGGL(sopvm_kernel<false>, dim3(waves), dim3(64), 0, 0, d_out, d_init, d_prog, steps, d_cyc, K);
    };
    launch();
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 5; rep++) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long cyc[8];
    hipMemcpy(cyc, d_cyc, sizeof cyc, hipMemcpyDeviceToHost);
    printf("%s %5d wavefronts x %d steps: %8.3f ms per launch = %6.2f us per step; wave 0: %lld cycles per step, levels (1,4,8,1,4,8 terms): %lld %lld %lld %lld %lld %lld\n",
           in_lds ? "tables in LDS   " : "tables in global", waves, steps, ms / 5, ms / 5 / steps * 1e3, cyc[0] / steps, cyc[1] / steps, cyc[2] / steps, cyc[3] / steps, cyc[4] / steps, cyc[5] / steps, cyc[6] / steps);
  }
  return 0;
}
