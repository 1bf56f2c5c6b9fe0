// Decision logic of mi_sinkhorn_dots' self-tuned stream schedule -- HIP-free, so it is unit-tested on the CPU with
// injected timings (tests/test_host_and_abi.py through mi_debug_tuner_script of the debug library).
//
// Which streams the two half-batches of a >= 64-pair call go to decides how well they overlap, and that depends on how
// the runtime mapped the process's streams onto the device's hardware queues (sinkhorn_dots.hip, ForkJoin).  So a caller
// stream TRIES the three schedules and keeps the fastest.  The rules, each the answer to a review finding of round 3:
//   * a decision belongs to ONE SHAPE (batch, n, m, iterations): a 64-pair and a 448-pair call sit in different regimes
//     (launch-gap- vs bandwidth-bound), so their trials are never compared; up to SK_SHAPES shapes are tuned per caller
//     stream, a further shape evicts the least recently used one that has no trial in flight (none free: that call runs
//     schedule SK_FALLBACK untried);
//   * SK_SAMPLES trials per schedule, round-robin, the MINIMUM per schedule decides: one noisy neighbour during the
//     window cannot pin a slow schedule unless it covers every sample of the fastest one;
//   * the decision EXPIRES: after SK_RETUNE_CALLS calls on it a new window is opened (the old decision stays in force for
//     the calls between trials being issued and harvested);
//   * a trial is a slot that is begun, then either finished with a time or ABANDONED (an error between its two events,
//     an event that could not be recorded): every begun slot ends, so a window always closes;
//   * a caller can PIN a schedule for every shape (mi_sinkhorn_dots_set_schedule) and read the decision in force
//     (mi_sinkhorn_dots_schedule); while pinned nothing is tried.
#pragma once

namespace mi {

constexpr int SK_SCHEDULES = 3;          // 0: halves on {caller, helper 0}; 1: halves on {helper 0, helper 1}; 2: unsplit
constexpr int SK_SAMPLES = 3;
constexpr int SK_TRIALS = SK_SAMPLES * SK_SCHEDULES;
constexpr int SK_SHAPES = 4;
constexpr int SK_FALLBACK = 2;           // schedule of a call that is neither tuned nor decided and must not try: unsplit
constexpr int SK_EAGER_DEFAULT = 0;      // eager calls of a window whose trials are all issued but not yet harvested
constexpr long long SK_RETUNE_CALLS = 8192;

struct TunerShape {
  int batch = 0, n = 0, m = 0, iterations = 0;
  bool operator==(const TunerShape &o) const {
    return batch == o.batch && n == o.n && m == o.m && iterations == o.iterations;
  }
};

struct TunerEntry {
  TunerShape shape;
  bool used = false;
  bool tuning = false;                   // a trial window is open
  int started = 0, ended = 0;            // slots begun / finished-or-abandoned in the open window
  bool open[SK_TRIALS] = {};             // begun and not yet ended
  double best[SK_SCHEDULES] = {};
  bool sampled[SK_SCHEDULES] = {};
  int decided = -1;                      // the decision in force (-1: none yet)
  long long calls_on_decision = 0;
  long long last_use = 0;
  void open_window() {
    tuning = true;
    started = ended = 0;
    for (int i = 0; i < SK_TRIALS; ++i) open[i] = false;
    for (int c = 0; c < SK_SCHEDULES; ++c) { best[c] = 0.0; sampled[c] = false; }
  }
  bool in_flight() const { return tuning && ended < started; }
};

struct TunerLogic {
  TunerEntry e[SK_SHAPES];
  long long clock = 0;
  int pinned = -1;

  int find(const TunerShape &s) const {
    for (int i = 0; i < SK_SHAPES; ++i)
      if (e[i].used && e[i].shape == s) return i;
    return -1;
  }
  // The decision in force for `s`: the pin, the shape's decision, or -1.
  int current(const TunerShape &s) const {
    if (pinned >= 0) return pinned;
    const int i = find(s);
    return i >= 0 ? e[i].decided : -1;
  }
  // A call inside a stream capture: nothing may be tried or recorded.
  int for_capture(const TunerShape &s) const {
    const int c = current(s);
    return c >= 0 ? c : SK_FALLBACK;
  }
  // An eager call.  Returns its schedule; *entry / *slot name the trial that brackets it (-1 / -1: none).
  int begin(const TunerShape &s, int *entry, int *slot) {
    *entry = *slot = -1;
    if (pinned >= 0) return pinned;
    ++clock;
    int i = find(s);
    if (i < 0) {
      int victim = -1;
      for (int k = 0; k < SK_SHAPES; ++k) {
        if (!e[k].used) { victim = k; break; }
        if (e[k].in_flight()) continue;
        if (victim < 0 || e[k].last_use < e[victim].last_use) victim = k;
      }
      if (victim < 0) return SK_FALLBACK;               // every entry has a trial in flight
      e[victim] = TunerEntry();
      e[victim].used = true;
      e[victim].shape = s;
      e[victim].open_window();
      i = victim;
    }
    TunerEntry &t = e[i];
    t.last_use = clock;
    if (!t.tuning && t.decided >= 0 && ++t.calls_on_decision >= SK_RETUNE_CALLS) t.open_window();
    if (t.tuning && t.started < SK_TRIALS) {
      const int sl = t.started++;
      t.open[sl] = true;
      *entry = i;
      *slot = sl;
      return sl % SK_SCHEDULES;
    }
    return t.decided >= 0 ? t.decided : SK_EAGER_DEFAULT;
  }
  void end_slot(int entry, int slot) {
    TunerEntry &t = e[entry];
    if (!t.used || !t.tuning || slot < 0 || slot >= SK_TRIALS || !t.open[slot]) return;
    t.open[slot] = false;
    if (++t.ended < SK_TRIALS) return;
    // the window closes: the fastest sampled schedule; nothing sampled at all (every trial abandoned) keeps the
    // decision in force, or the eager default
    int pick = -1;
    for (int c = 0; c < SK_SCHEDULES; ++c)
      if (t.sampled[c] && (pick < 0 || t.best[c] < t.best[pick])) pick = c;
    if (pick < 0) pick = t.decided >= 0 ? t.decided : SK_EAGER_DEFAULT;
    t.decided = pick;
    t.tuning = false;
    t.calls_on_decision = 0;
  }
  void finish(int entry, int slot, double ms) {
    if (entry < 0 || entry >= SK_SHAPES) return;
    TunerEntry &t = e[entry];
    if (t.used && t.tuning && slot >= 0 && slot < SK_TRIALS && t.open[slot] && ms > 0.0) {
      const int c = slot % SK_SCHEDULES;
      if (!t.sampled[c] || ms < t.best[c]) t.best[c] = ms;
      t.sampled[c] = true;
    }
    end_slot(entry, slot);
  }
  void abandon(int entry, int slot) {
    if (entry < 0 || entry >= SK_SHAPES) return;
    end_slot(entry, slot);
  }
  // pin (0 .. SK_SCHEDULES - 1) or unpin + forget every decision (-1)
  bool set(int schedule) {
    if (schedule < -1 || schedule >= SK_SCHEDULES) return false;
    pinned = schedule;
    if (schedule < 0)
      for (int k = 0; k < SK_SHAPES; ++k)
        if (!e[k].in_flight()) e[k] = TunerEntry();
    return true;
  }
};

}  // namespace mi
