// CPU dry test of the multi-process stop protocol (clipper_amd/csrc/host_batch.hpp): R simulated
// ranks — one host thread each, running the REAL loop template — with their own in-order "stream"
// (a queue of closures executed when the host waits) share a collective that completes only when
// EVERY rank has entered it. The device state converges after a given number of iterations.
// Checks: no rank hangs in a collective its peers never queue, every rank queues the same number
// of iterations, every collective is entered by all ranks, the ranks stop within two batches of
// convergence — for 1, 2, 3 and 8 ranks, batch sizes 1..7, convergence on and off batch edges.
//   g++ -std=c++17 -O1 -pthread -I clipper_amd/csrc tests/cpp/test_batch_protocol.cpp -o /tmp/t && /tmp/t
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "host_batch.hpp"

// One simulated rank: an in-order stream of queued operations, executed when the host waits.
// The collective of iteration i completes only when EVERY rank has entered collective i — a rank
// whose peers never queue it blocks until the deadline: that is the hang this protocol prevents.
struct Collective {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<int> arrived;
  int R;
  bool enter(int64_t idx) {
    std::unique_lock<std::mutex> lk(mu);
    if (static_cast<int64_t>(arrived.size()) <= idx) arrived.resize(idx + 1, 0);
    ++arrived[idx];
    cv.notify_all();
    return cv.wait_for(lk, std::chrono::seconds(5), [&] { return arrived[idx] >= R; });
  }
};

struct Rank {
  std::deque<std::function<bool()>> stream;
  int64_t iters_run = 0;  // device-side: iterations that did work
  int done = 0;           // SolveShared::done on the device
  int host_done[2] = {0, 0};
  bool drain() {  // execute everything queued so far, in order
    while (!stream.empty()) {
      if (!stream.front()()) return false;
      stream.pop_front();
    }
    return true;
  }
};

static int simulate(int R, int batch, int64_t converge_at) {
  Collective coll;
  coll.R = R;
  std::vector<Rank> ranks(R);
  std::vector<int64_t> queued(R, 0);
  std::vector<int> rcs(R, 0);
  std::vector<std::thread> th;
  for (int k = 0; k < R; ++k)
    th.emplace_back([&, k]() {
      Rank& rk = ranks[k];
      int64_t next = 0;
      rcs[k] = clipper_hip::run_batched_until_done(
          batch,
          [&]() {  // one iteration: the kernels (no-ops once done), then the collective
            const int64_t idx = next++;
            rk.stream.push_back([&rk, &coll, idx, converge_at]() {
              if (!coll.enter(idx)) return false;  // peers never came: hang
              if (!rk.done && ++rk.iters_run >= converge_at) rk.done = 1;
              return true;
            });
            return 0;
          },
          [&](int slot) {
            rk.stream.push_back([&rk, slot]() { rk.host_done[slot] = rk.done; return true; });
            return 0;
          },
          [&](int slot, bool& d) {
            if (!rk.drain()) return -1;  // "the stream never completed"
            d = rk.host_done[slot] != 0;
            return 0;
          },
          &queued[k]);
      if (rcs[k] == 0 && !rk.drain()) rcs[k] = -2;  // what was queued past the last snapshot
    });
  for (auto& t : th) t.join();
  for (int k = 0; k < R; ++k)
    if (rcs[k]) { printf("R=%d batch=%d converge=%lld: rank %d hung (rc %d)\n", R, batch, (long long)converge_at, k, rcs[k]); return 1; }
  for (int k = 1; k < R; ++k)
    if (queued[k] != queued[0]) { printf("ranks queued %lld vs %lld iterations\n", (long long)queued[0], (long long)queued[k]); return 1; }
  for (size_t i = 0; i < coll.arrived.size(); ++i)
    if (coll.arrived[i] != R) { printf("collective %zu entered by %d of %d ranks\n", i, coll.arrived[i], R); return 1; }
  if (static_cast<int64_t>(coll.arrived.size()) != queued[0]) { printf("collectives != iterations\n"); return 1; }
  if (queued[0] % batch != 0 || queued[0] < converge_at || queued[0] > converge_at + 2 * batch) {
    printf("queued %lld iterations for convergence at %lld, batch %d\n", (long long)queued[0], (long long)converge_at, batch);
    return 1;
  }
  return 0;
}

// the real template against a single simulated device: the snapshot that reports `done` is read
// one batch late by construction; the iteration count is a multiple of the batch
static int drive_template(int batch, int64_t converge_at) {
  int64_t dev_iters = 0, n = 0;
  int dev_done = 0, host[2] = {0, 0};
  int rc = clipper_hip::run_batched_until_done(
      batch,
      [&]() { if (!dev_done && ++dev_iters >= converge_at) dev_done = 1; return 0; },
      [&](int slot) { host[slot] = dev_done; return 0; },
      [&](int slot, bool& d) { d = host[slot] != 0; return 0; }, &n);
  if (rc) return 1;
  if (n % batch != 0 || n < converge_at || n > converge_at + 2 * batch) {
    printf("template: %lld iterations for convergence at %lld, batch %d\n", (long long)n, (long long)converge_at, batch);
    return 1;
  }
  // an error from enqueue stops at once
  int calls = 0;
  rc = clipper_hip::run_batched_until_done(
      batch, [&]() { return ++calls == 3 ? -7 : 0; }, [&](int) { return 0; },
      [&](int, bool& d) { d = false; return 0; }, &n);
  return (rc == -7 && calls == 3) ? 0 : 1;
}

// The hold-aware loop with R simulated ranks: the device goes on hold at iteration `hold_at` (every
// iteration behind it does nothing but still enters its collective), every rank must notice from the
// same snapshot, drain, "build" and resume; then the solve converges `more` real iterations later.
static int simulate_holds(int R, int batch, int64_t hold_at, int64_t more) {
  Collective coll;
  coll.R = R;
  std::vector<Rank> ranks(static_cast<size_t>(R));
  std::vector<int> rcs(static_cast<size_t>(R), 0), holds(static_cast<size_t>(R), 0);
  std::vector<int64_t> queued(static_cast<size_t>(R), 0), real(static_cast<size_t>(R), 0);
  std::vector<std::thread> th;
  for (int k = 0; k < R; ++k)
    th.emplace_back([&, k]() {
      Rank& rk = ranks[k];
      int64_t next = 0;
      int held = 0;        // device state: 0 running, 1 on hold
      bool was_held = false;
      int host_state[2] = {0, 0};
      rcs[k] = clipper_hip::run_batched_with_holds(
          batch,
          [&]() {
            const int64_t idx = next++;
            rk.stream.push_back([&, idx]() {
              if (!coll.enter(idx)) return false;
              if (rk.done || held) return true;  // nothing runs on hold or past the end
              ++real[k];
              if (!was_held && real[k] == hold_at) held = 1;
              else if (was_held && real[k] >= hold_at + more) rk.done = 1;
              return true;
            });
            return 0;
          },
          [&](int slot) {
            rk.stream.push_back([&, slot]() { host_state[slot] = rk.done ? 1 : (held ? 2 : 0); return true; });
            return 0;
          },
          [&](int slot, int& st) {
            if (!rk.drain()) return -1;
            st = host_state[slot];
            return 0;
          },
          [&]() {
            if (!rk.drain()) return -3;
            held = 0;
            was_held = true;
            ++holds[k];
            return 0;
          },
          &queued[k]);
      if (rcs[k] == 0 && !rk.drain()) rcs[k] = -2;
    });
  for (auto& t : th) t.join();
  for (int k = 0; k < R; ++k)
    if (rcs[k] || holds[k] != 1 || real[k] != hold_at + more) {
      printf("holds R=%d batch=%d hold_at=%lld: rank %d rc %d holds %d real %lld\n", R, batch, (long long)hold_at, k, rcs[k],
             holds[k], (long long)real[k]);
      return 1;
    }
  for (int k = 1; k < R; ++k)
    if (queued[k] != queued[0]) { printf("holds: ranks queued %lld vs %lld\n", (long long)queued[0], (long long)queued[k]); return 1; }
  for (size_t i = 0; i < coll.arrived.size(); ++i)
    if (coll.arrived[i] != R) { printf("holds: collective %zu entered by %d of %d ranks\n", i, coll.arrived[i], R); return 1; }
  return 0;
}

int main() {
  int bad = 0;
  for (int R : {1, 2, 3, 8})
    for (int batch = 1; batch <= 5; ++batch)
      for (int64_t hold_at : {1, 2, 4, 5, 9, 17})
        for (int64_t more : {1, 3, 8})
          bad += simulate_holds(R, batch, hold_at, more);
  for (int R : {1, 2, 3, 8})
    for (int batch = 1; batch <= 7; ++batch)
      for (int64_t c : {1, 2, 3, 4, 5, 8, 9, 16, 17, 31, 100})
        bad += simulate(R, batch, c);
  for (int batch = 1; batch <= 7; ++batch)
    for (int64_t c : {1, 2, 4, 5, 8, 9, 33, 100})
      bad += drive_template(batch, c);
  printf(bad ? "FAILED (%d)\n" : "batch protocol ok\n", bad);
  return bad ? 1 : 0;
}
