// Host-side schedule of the fast BAD kernel's LDS gathers (bad.hip, bad_fast_kernel) -- plain C++, no HIP, so it is
// unit-testable on the CPU (mi_debug_bad_plan_passes, tests/test_host_and_abi.py).
//
// The kernel evaluates 64 pairs per round (lane l = pair), eight table reads per pair: the four corners that enter
// s1 - s2 with + and the four with -.  Every read instruction is 64 scattered 4-byte LDS addresses, served one
// half-wave (32 lanes) at a time in as many passes as the most loaded of the 32 banks has DISTINCT words.  The
// addresses are the same for every keypoint, so the schedule is chosen once per pair table.  Two freedoms:
//   1. inside a sign class the order in which a lane reads its four corners (free: the sum commutes);
//   2. the ROUND in which a lane evaluates each of its pairs: lane l owns the pairs l, l + 64, ... of the canonical
//      order and may take them in any order, because the kernel rebuilds the canonical words from the rounds' ballots
//      with one 64-bit mask per (round, canonical group) -- scalar work -- so the packed descriptor does not change.
// A unit = (round, half-wave, sign class): 32 lanes x 4 reads.  However the reads are ordered, a unit needs at least as
// many passes as its most loaded bank has distinct words (every one of them has to be served in some instruction), and
// a good order gets close to that bound -- so step A anneals the lanes' round assignment towards small bank maxima,
// step B anneals the corner order of every unit.  On the reference's 512-pair table: 420 passes as the table stands,
// 328 with a greedy corner order (round 1 of this project; bound 243 with the rounds as they stand), bound 156 and 184
// reached with both steps (256-pair table: 211 -> 101); 128 (64) would be conflict-free.
// Deterministic (fixed-seed LCG): the plan is a function of the table.
#pragma once

#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <vector>

#ifndef MI_BAD_PLAN_ORDER_ITERS
#define MI_BAD_PLAN_ORDER_ITERS 300000
#endif

namespace mi {

struct BadGatherSchedule {
  int num_pairs = 0;
  std::vector<int> exec_pair;          // [round * 64 + lane] -> canonical pair index (always lane + 64 * something)
  std::vector<uint16_t> pos, neg;      // [(round * 64 + lane) * 4 + read] -> word index into the table
  int passes = 0;                      // LDS passes per keypoint of this schedule (the conflict-free minimum is num_pairs / 4)
  int passes_canonical = 0;            // ... of the table as it stands (rounds and corner order untouched)
};

namespace bad_plan_detail {

struct Lcg {
  uint32_t s;
  explicit Lcg(uint32_t seed) : s(seed) {}
  uint32_t next() { s = s * 1664525u + 1013904223u; return s >> 8; }
  int below(int n) { return (int)(next() % (uint32_t)n); }
  double unit() { return (double)next() / 16777216.0; }
};

// distinct words per bank of `count` words
struct BankLoad {
  uint16_t words[32][32];
  int n[32];
  void clear() { for (int b = 0; b < 32; ++b) n[b] = 0; }
  void add(uint16_t w) {
    const int b = w & 31;
    for (int k = 0; k < n[b]; ++k)
      if (words[b][k] == w) return;
    if (n[b] < 32) words[b][n[b]++] = w;
  }
  int worst() const { int m = 0; for (int b = 0; b < 32; ++b) m = n[b] > m ? n[b] : m; return m; }
  int squares() const { int s = 0; for (int b = 0; b < 32; ++b) s += n[b] * n[b]; return s; }
};

// passes of one read instruction over 64 lanes (two half-waves), `words[lane * stride]`
inline int instruction_passes(const uint16_t *words, int stride) {
  int total = 0;
  BankLoad load;
  for (int half = 0; half < 2; ++half) {
    load.clear();
    for (int l = 32 * half; l < 32 * half + 32; ++l) load.add(words[l * stride]);
    total += load.worst();
  }
  return total;
}

}  // namespace bad_plan_detail

// corners[(pair * 2 + sign) * 4 + q]: table word indices of pair `pair` (sign 0: the four + corners, 1: the four -)
inline BadGatherSchedule schedule_bad_gathers(const std::vector<uint16_t> &corners, int num_pairs) {
  using namespace bad_plan_detail;
  BadGatherSchedule out;
  out.num_pairs = num_pairs;
  const int rounds = num_pairs / 64;
  out.exec_pair.resize(num_pairs);
  for (int i = 0; i < num_pairs; ++i) out.exec_pair[i] = i;
  auto corner = [&](int pair, int sign, int q) { return corners[(size_t)(pair * 2 + sign) * 4 + q]; };

  auto total_passes = [&](const std::vector<uint16_t> &pos, const std::vector<uint16_t> &neg) {
    int t = 0;
    for (int g = 0; g < rounds; ++g)
      for (int q = 0; q < 4; ++q)
        t += instruction_passes(&pos[(size_t)g * 256 + q], 4) + instruction_passes(&neg[(size_t)g * 256 + q], 4);
    return t;
  };
  auto fill = [&](std::vector<uint16_t> &pos, std::vector<uint16_t> &neg) {
    pos.resize((size_t)num_pairs * 4);
    neg.resize((size_t)num_pairs * 4);
    for (int i = 0; i < num_pairs; ++i)
      for (int q = 0; q < 4; ++q) {
        pos[(size_t)i * 4 + q] = corner(out.exec_pair[i], 0, q);
        neg[(size_t)i * 4 + q] = corner(out.exec_pair[i], 1, q);
      }
  };
  fill(out.pos, out.neg);
  out.passes_canonical = total_passes(out.pos, out.neg);

  // ---- step A: which round a lane evaluates each of its pairs in.  Energy of a (round, half) = over both sign
  // classes, 100 x the largest number of distinct words in a bank + the sum of squares (a smooth tie-break).
  Lcg rng(12345u);
  if (rounds > 1) {
    auto energy = [&](int g, int half) {
      int e = 0;
      BankLoad load;
      for (int sign = 0; sign < 2; ++sign) {
        load.clear();
        for (int l = 32 * half; l < 32 * half + 32; ++l)
          for (int q = 0; q < 4; ++q) load.add(corner(out.exec_pair[g * 64 + l], sign, q));
        e += 100 * load.worst() + load.squares();
      }
      return e;
    };
    std::vector<int> e((size_t)rounds * 2);
    for (int g = 0; g < rounds; ++g)
      for (int half = 0; half < 2; ++half) e[g * 2 + half] = energy(g, half);
    const int iters = 6000 * rounds;
    for (int it = 0; it < iters; ++it) {
      const int l = rng.below(64), half = l >> 5;
      const int g1 = rng.below(rounds);
      int g2 = rng.below(rounds - 1);
      if (g2 >= g1) ++g2;
      std::swap(out.exec_pair[g1 * 64 + l], out.exec_pair[g2 * 64 + l]);
      const int n1 = energy(g1, half), n2 = energy(g2, half);
      const int before = e[g1 * 2 + half] + e[g2 * 2 + half], after = n1 + n2;
      const double temp = std::max(0.01, 40.0 * (1.0 - (double)it / iters));
      if (after <= before || rng.unit() < std::exp((double)(before - after) / temp)) {
        e[g1 * 2 + half] = n1;
        e[g2 * 2 + half] = n2;
      } else {
        std::swap(out.exec_pair[g1 * 64 + l], out.exec_pair[g2 * 64 + l]);
      }
    }
  }
  fill(out.pos, out.neg);

  // ---- step B: corner order per (round, half, sign): 32 lanes x 4 reads, energy = 100 x (passes of the four
  // instructions) + squares of the bank loads, updated incrementally (a move swaps two reads of one lane: four
  // (instruction, word) counters change); the best state seen is kept
  struct Unit {
    uint8_t cnt[4][2048];        // lanes reading this word in instruction q
    int distinct[4][32];         // distinct words per bank
    int hist[4][34];             // banks per load
    int worst[4];
    int squares;
    void add(int q, uint16_t w) {
      if (cnt[q][w]++ != 0) return;
      const int v = distinct[q][w & 31]++;
      --hist[q][v];
      ++hist[q][v + 1];
      if (v + 1 > worst[q]) worst[q] = v + 1;
      squares += 2 * v + 1;
    }
    void remove(int q, uint16_t w) {
      if (--cnt[q][w] != 0) return;
      const int v = distinct[q][w & 31]--;
      --hist[q][v];
      ++hist[q][v - 1];
      if (hist[q][worst[q]] == 0) --worst[q];
      squares -= 2 * v - 1;
    }
    int energy() const { return 100 * (worst[0] + worst[1] + worst[2] + worst[3]) + squares; }
  };
  std::vector<Unit> unit_store(1);
  Unit &u = unit_store[0];
  for (int g = 0; g < rounds; ++g)
    for (int half = 0; half < 2; ++half)
      for (int sign = 0; sign < 2; ++sign) {
        uint16_t *w = (sign ? out.neg.data() : out.pos.data()) + ((size_t)g * 64 + 32 * half) * 4;   // [32][4]
        u = Unit();
        for (int q = 0; q < 4; ++q) u.hist[q][0] = 32;
        for (int l = 0; l < 32; ++l)
          for (int q = 0; q < 4; ++q) u.add(q, w[l * 4 + q] & 2047);
        int cur = u.energy();
        uint16_t best[128];
        std::copy(w, w + 128, best);
        int best_e = cur;
        const int iters = MI_BAD_PLAN_ORDER_ITERS;
        for (int it = 0; it < iters; ++it) {
          const int l = rng.below(32), a = rng.below(4);
          int b = rng.below(3);
          if (b >= a) ++b;
          const uint16_t wa = w[l * 4 + a], wb = w[l * 4 + b];
          if (wa == wb) continue;
          u.remove(a, wa); u.remove(b, wb); u.add(a, wb); u.add(b, wa);
          const int after = u.energy();
          const double temp = std::max(0.01, 30.0 * (1.0 - (double)it / iters));
          if (after <= cur || rng.unit() < std::exp((double)(cur - after) / temp)) {
            std::swap(w[l * 4 + a], w[l * 4 + b]);
            cur = after;
            if (cur < best_e) {
              best_e = cur;
              std::copy(w, w + 128, best);
            }
          } else {
            u.remove(a, wb); u.remove(b, wa); u.add(a, wa); u.add(b, wb);
          }
        }
        std::copy(best, best + 128, w);
      }
  out.passes = total_passes(out.pos, out.neg);
  return out;
}

}  // namespace mi
