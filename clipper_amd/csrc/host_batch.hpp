// host_batch.hpp — the stop protocol of a multi-process solve (no HIP in here: tests/cpp/
// test_batch_protocol.cpp drives it with a simulated device).
//
// Every rank must queue the SAME number of solver iterations: each iteration of a column shard
// holds a collective (the all-gather of the pass's block), and a rank that stopped queueing
// one iteration earlier than its peers would leave them waiting in it forever. No rank may
// therefore decide to stop on anything but data that is bit-identical on all ranks: the
// SolveShared record (every rank recomputes all O(m) work on the same gathered bits). The loop
// queues iterations in batches; after batch n it queues a snapshot of the record (an async
// copy + an event) and, BEFORE looking at it, queues batch n + 1 — only then does it wait for
// snapshot n. The device is never idle, and all ranks read `done` from the same snapshot index,
// hence stop after the same batch. Iterations queued past convergence are no-ops on the device
// but still perform their exchange (the kernels exit on `done`, the collective does not know).
#pragma once

#include <cstdint>

namespace clipper_hip {

// enqueue()            queue one solver iteration (incl. its exchange); 0 or an error code
// snapshot(slot)       queue a copy of the shared record into host slot `slot` (0 | 1); 0 or error
// wait_done(slot, d)   wait for that copy, d = its `done` flag; 0 or error
// Returns 0 or the first error; *iterations = how many iterations were queued.
template <class Enqueue, class Snapshot, class WaitDone>
int run_batched_until_done(int batch, Enqueue&& enqueue, Snapshot&& snapshot, WaitDone&& wait_done,
                           int64_t* iterations) {
  int slot = 0;
  bool have_prev = false, done = false;
  int64_t n = 0;
  if (batch < 1) batch = 1;
  while (!done) {
    for (int it = 0; it < batch; ++it) {
      if (int rc = enqueue()) return rc;
      ++n;
    }
    if (int rc = snapshot(slot)) return rc;
    if (have_prev) {
      bool d = false;
      if (int rc = wait_done(slot ^ 1, d)) return rc;
      done = d;
    }
    have_prev = true;
    slot ^= 1;
  }
  if (iterations) *iterations = n;
  return 0;
}

// The same loop for a solve that may go on HOLD (k_solver.hip.h, LIVE ROWS): wait_state(slot, st) reports
// 0 = running, 1 = done, 2 = on hold — from the same snapshot index on every rank. On hold every rank has
// queued the same iterations (the ones behind the hold do nothing but still perform their exchange), so
// every rank can drain its stream without waiting for a peer that queued less; on_hold() does that, builds
// the view and lifts the hold; the snapshots taken before are stale and the loop primes itself again.
template <class Enqueue, class Snapshot, class WaitState, class OnHold>
int run_batched_with_holds(int batch, Enqueue&& enqueue, Snapshot&& snapshot, WaitState&& wait_state,
                           OnHold&& on_hold, int64_t* iterations) {
  int slot = 0;
  bool have_prev = false, done = false;
  int64_t n = 0;
  if (batch < 1) batch = 1;
  while (!done) {
    for (int it = 0; it < batch; ++it) {
      if (int rc = enqueue()) return rc;
      ++n;
    }
    if (int rc = snapshot(slot)) return rc;
    if (have_prev) {
      int st = 0;
      if (int rc = wait_state(slot ^ 1, st)) return rc;
      if (st == 2) {
        if (int rc = on_hold()) return rc;
        have_prev = false;
        slot ^= 1;
        continue;
      }
      done = (st == 1);
    }
    have_prev = true;
    slot ^= 1;
  }
  if (iterations) *iterations = n;
  return 0;
}

}  // namespace clipper_hip
